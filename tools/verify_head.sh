set -x
mkdir -p gpurun_out
{
echo "HEAD $(cat tools/.head 2>/dev/null)"
date
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "--- smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5
echo "--- bench (driver command)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -3
date
} > gpurun_out/r06_verify_head.txt 2>&1
tail -30 gpurun_out/r06_verify_head.txt
