( time timeout 900 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err ) 2> gpurun_out/r3_bench.time; tail -3 gpurun_out/r3_bench.time
