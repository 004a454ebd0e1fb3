mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_lr.py -x -q 2>&1 | tail -5 > gpurun_out/r6/lr_tests.txt
python -m pytest tests/test_gpu_net.py tests/test_gpu_00_dp.py -x -q -k "not 256" 2>&1 | tail -5 > gpurun_out/r6/lr_net_tests.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --breakdown --detail mny_lr_prep,mny_lr_gram,mny_pw_lr_fix,mny_lr_wfix > gpurun_out/r6/lr_c1.json 2> gpurun_out/r6/lr_c1.txt
bash tools/rounds/r5_ab.sh "MNY_NO_LR=1 MNY_NO_LR=0 MNY_LR_PREP_MAIN=1" c1 > gpurun_out/r6/lr_ab.txt 2>&1
