set -e
mkdir -p gpurun_out/p6o
cd /root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "step or raw or mask or sis or smoke" > gpurun_out/p6o/pytest.log 2>&1 || { tail -30 gpurun_out/p6o/pytest.log; exit 1; }
tail -3 gpurun_out/p6o/pytest.log
python tools/ab_libs.py 3 "" libglb_base.so > gpurun_out/p6o/ab_libs.log 2>&1
cat gpurun_out/p6o/ab_libs.log
