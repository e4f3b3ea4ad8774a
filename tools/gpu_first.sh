set -x
rocminfo | grep -E "gfx|Marketing" | head -4
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40
