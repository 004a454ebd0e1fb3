python -m pytest tests/test_gpu_exdw.py -m gpu -x -q 2>&1 | tail -12
python tools/bench_exdw.py fwd 256 2>&1 | grep -v amdgpu
MNY_EXDW_STATS=direct python tools/bench_exdw.py fwd 256 2>&1 | grep -v amdgpu
