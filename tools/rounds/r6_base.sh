# round-6 starting state: plan dumps (launch order, shapes, routes) + per-call breakdown of both configs + un-bracketed A/B baselines
mkdir -p gpurun_out/r6
python tools/dump_plan.py mbv2 256 352 f32 > gpurun_out/r6/plan_c1.txt 2>gpurun_out/r6/plan_c1.err
python tools/dump_plan.py mbv3 64 512 bf16 > gpurun_out/r6/plan_c3.txt 2>gpurun_out/r6/plan_c3.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --breakdown --detail all > gpurun_out/r6/base_c1.json 2> gpurun_out/r6/base_c1.txt
python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 10 --warmup 4 --no-cpu-baseline --no-nms --breakdown --detail all > gpurun_out/r6/base_c3.json 2> gpurun_out/r6/base_c3.txt
bash tools/rounds/r5_ab.sh "X=0" both > gpurun_out/r6/base_ab.txt 2>&1
