# the driver's own commands at the final commit: smoke(), the default bench line (compact stdout, verbose line on stderr), configs[3] standalone lines
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6/bench_n1.json 2> gpurun_out/r6/bench_n1.err
python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --no-nms --no-cpu-baseline > gpurun_out/r6/bench_mbv3_512_bf16.json 2> gpurun_out/r6/bench_mbv3_512_bf16.err
python bench.py --arch mbv3 --size 512 --batch 64 --dtype f32 --no-nms --no-cpu-baseline > gpurun_out/r6/bench_mbv3_512_fp32.json 2> gpurun_out/r6/bench_mbv3_512_fp32.err
