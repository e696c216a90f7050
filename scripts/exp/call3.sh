mkdir -p gpurun_out/c3
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -25 > gpurun_out/c3/pytest_gpu.log
cat gpurun_out/c3/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
