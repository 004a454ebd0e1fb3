R=$GRAFT_REPO_ROOT/gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail mny_dw_fwd"
$B > /dev/null 2> $R/dw_A.txt
MNY_DW_RES=512 $B > /dev/null 2> $R/dw_H.txt
MNY_DW_RES=640 $B > /dev/null 2> $R/dw_I.txt
MNY_DW_RES=384 $B > /dev/null 2> $R/dw_J.txt
MNY_DW_RES=768 MNY_DW_TH=32 $B > /dev/null 2> $R/dw_K.txt
MNY_DW_RES=768 MNY_DW_XCD=0 $B > /dev/null 2> $R/dw_L.txt
MNY_DW_RES=768 MNY_DW_NT=0 $B > /dev/null 2> $R/dw_M.txt
for f in A H I J K L M; do echo "== $f: $(grep '^mny_dw_fwd ' $R/dw_$f.txt) | $(grep wall $R/dw_$f.txt)"; done
