cd /root/repo
O=gpurun_out/r5_23; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2 3; do timeout 400 $B --workload msm > $O/msm_$rep.json 2> $O/msm_$rep.err; done
python exp/bench_lines.py $O
