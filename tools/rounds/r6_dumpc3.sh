mkdir -p gpurun_out/r6
python tools/dump_plan.py mbv3 64 512 bf16 > gpurun_out/r6/plan_c3.txt 2>&1
