for shape in "123904 96 576" "123904 96 384" "123904 96 512" "123904 64 384" "123904 80 512" "30976 80 1024"; do
  for kind in pw pwx; do
    echo "== $kind $shape"
    for env in "X=1" "MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_w1.so" "X=2" "MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_w1.so"; do
      echo -n "${env:0:10}: "; env $env python tools/kbench.py $kind $shape 50 2>&1 | tail -1
    done
  done
done
