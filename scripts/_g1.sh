mkdir -p gpurun_out/g1
export PYTHONPATH=.
timeout 200 python scripts/gpu_mlp_marks.py > gpurun_out/g1/marks.txt 2>&1
head -14 gpurun_out/g1/marks.txt | cut -c1-230
timeout 300 python bench.py --workload sac > gpurun_out/g1/bench_sac.txt 2>&1
tail -n 1 gpurun_out/g1/bench_sac.txt | cut -c1-200
