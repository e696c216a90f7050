mkdir -p gpurun_out/c2
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_dp4.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c2/pytest_api.log
cat gpurun_out/c2/pytest_api.log
for M in fp32 bf16; do
  timeout 300 python scripts/bench_api_train.py --conv-mode $M 2>/dev/null | grep '^{' > gpurun_out/c2/api_train_$M.json
  timeout 300 python scripts/bench_api_train.py --conv-mode $M --world 2 2>gpurun_out/c2/api_w2_$M.err | grep '^{' > gpurun_out/c2/api_train_${M}_world2.json
  ST_SHARE_GPU=1 ST_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 40 --warmup 5 --steps-only --conv-mode $M 2>gpurun_out/c2/bench_w2_$M.err | grep '^{' > gpurun_out/c2/bench_${M}_world2_shared.json
done
cat gpurun_out/c2/*.json
tail -5 gpurun_out/c2/api_w2_fp32.err
