bash tools/profile_round_2.sh
cd /root/repo
for i in 1 2 3; do python3 bench.py --workload api-coro --auto-kv --llm-gather --steps 20 --warmup 3 --no-cpu --step-times > gpurun_out/prof2/bench_api-coro_autokv_llmgather_run$i.json 2> gpurun_out/prof2/api-coro_autokv_llmgather_step_times_run$i.txt; done
tail -3 gpurun_out/prof2/api-coro_autokv_llmgather_step_times_run1.txt
