set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest1.log; tail -30 gpurun_out/pytest1.log
