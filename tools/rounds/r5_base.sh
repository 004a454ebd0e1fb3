# round-5 baseline: per-call tables of both configurations on one box
set -x
mkdir -p gpurun_out/r5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail all > gpurun_out/r5/base_c1.json 2> gpurun_out/r5/base_c1.txt
python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail all > gpurun_out/r5/base_c3.json 2> gpurun_out/r5/base_c3.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms > gpurun_out/r5/base_c1_plain.json 2> gpurun_out/r5/base_c1_plain.err
python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-nms > gpurun_out/r5/base_c3_plain.json 2> gpurun_out/r5/base_c3_plain.err
tail -3 gpurun_out/r5/base_c1.txt; cat gpurun_out/r5/base_c1_plain.json | head -c 600; echo; cat gpurun_out/r5/base_c3_plain.json | head -c 600
