# round 6: low-rank BN backward — kernel tests, whole-network oracle parity (bs 8 / 64), plan dump, same-box A/B against MNY_NO_LR=1
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_lr.py -x -q 2>&1 | tail -40 > gpurun_out/r6/lr_tests.txt
python -m pytest tests/test_gpu_net.py -x -q -k "not 256" 2>&1 | tail -30 > gpurun_out/r6/lr_net_tests.txt
python tools/dump_plan.py mbv2 256 352 f32 > gpurun_out/r6/plan_c1_lr.txt 2>gpurun_out/r6/plan_c1_lr.err
bash tools/rounds/r5_ab.sh "MNY_NO_LR=1,MNY_NO_ADDRED=1 MNY_NO_LR=1 MNY_NO_LR=0" c1 > gpurun_out/r6/lr_ab.txt 2>&1
