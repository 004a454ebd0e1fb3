# knob sweeps of the wide-output kernel through tools/kbench.py (same box); edit the lists as needed
for shape in "123904 64 384" "123904 96 384" "30976 96 1152"; do
  for kind in pw pwx; do
    echo "== $kind $shape"
    for env in "X=1" "MNY_WIDE_TNB=4" "MNY_WIDE_TNB=2" "X=2" "MNY_WIDE_TNB=4"; do
      echo -n "${env}: "; env $env python tools/kbench.py $kind $shape 50 2>&1 | tail -1
    done
  done
done
