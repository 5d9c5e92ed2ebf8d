set -e
mkdir -p gpurun_out/p6p
cd /root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/p6p/pytest.log 2>&1 || { tail -40 gpurun_out/p6p/pytest.log; exit 1; }
tail -3 gpurun_out/p6p/pytest.log
python bench.py --workload sis --per-row-masks --no-cpu > gpurun_out/p6p/bench_sis_rowmasks.json 2> gpurun_out/p6p/bench_sis_rowmasks.err
tail -c 1500 gpurun_out/p6p/bench_sis_rowmasks.json
