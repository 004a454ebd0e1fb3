python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "stem" 2>&1 | tail -2
R=$GRAFT_REPO_ROOT/gpurun_out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown > /dev/null 2> $R/ab_new_bd.txt
MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_prev.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown > /dev/null 2> $R/ab_prev_bd.txt
paste <(grep "ms/step" $R/ab_new_bd.txt | awk '{print $1,$2}') <(grep "ms/step" $R/ab_prev_bd.txt | awk '{print $1,$2}') | grep "stem\|exdw_bwd\|sum"
