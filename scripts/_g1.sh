mkdir -p gpurun_out/g1
export PYTHONPATH=.
timeout 200 python scripts/gpu_mlp_marks.py > gpurun_out/g1/marks.txt 2>&1
cat gpurun_out/g1/marks.txt
