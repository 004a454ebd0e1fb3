# mid-size GEMM shapes under the plan knobs (kbench pwx = forward with view transform + stats)
for shape in "123904 64 384" "123904 96 576" "30976 160 960" "123904 384 64" "123904 576 96" "30976 960 160"; do
  for env in "X=1" "MNY_NT_MAXTN=2" "MNY_NT_MAXTN=3" "MNY_NT_MAXTN=4" "MNY_NT_BALANCE_AI=10" "MNY_NT_BALANCE=0"; do
    echo -n "$env: "; env $env python tools/kbench.py pwx $shape 50
  done
done
