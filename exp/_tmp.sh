cd /root/repo
O=gpurun_out/r5_22; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "export" > $O/pytest_export.log 2>&1; echo "pytest export rc $?"; tail -3 $O/pytest_export.log
python bench.py --sub --suite main --workload msm --ring 1 --steps 3 --warmup 1 --latency-steps 0 --consumer-ready 3 --no-cpu-baseline --traffic off > $O/consumer.json 2> $O/consumer.err
grep -o '"consumer_ready_ms_per_step": [0-9.]*' $O/consumer.json | head -1
python exp/next_rows_bench.py > $O/next_rows.json 2> $O/next_rows.err; tail -c 1500 $O/next_rows.json
