#!/bin/bash
# Round profile set (run on the GPU box from the repo root): usage  bash tools/collect_profiles.sh r02
# kernel stats of the headline step and of --workload long, the FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes
# (each its own rocprofv3 run), every SURVEY 8d workload line, and the full bench line with cpu_baseline.
# Afterwards, in the container:  bash tools/collect_profiles.sh r02 copy   moves the summaries into profiles/.
tag=$1
if [ "$2" = copy ]; then
  cp gpurun_out/prof_${tag}w/${tag}w_kernel_stats.csv profiles/${tag}_kernel_stats.csv
  cp gpurun_out/prof_${tag}l/${tag}l_kernel_stats.csv profiles/${tag}_long_kernel_stats.csv
  python tools/summ.py profiles/${tag}_kernel_stats.csv 6 60 > profiles/${tag}_kernel_stats_summary.txt
  python tools/summ.py profiles/${tag}_long_kernel_stats.csv 3 40 > profiles/${tag}_long_kernel_stats_summary.txt
  cp gpurun_out/pmc_${tag}_fetch.csv profiles/${tag}_pmc_fetch_size.csv
  cp gpurun_out/pmc_${tag}_write.csv profiles/${tag}_pmc_write_size.csv
  cp gpurun_out/pmc_${tag}_mfma.csv profiles/${tag}_pmc_mfma_busy.csv
  rm -f profiles/${tag}_pmc_traffic.json
  python tools/pmc_traffic.py wgan profiles/${tag}_pmc_fetch_size.csv profiles/${tag}_pmc_write_size.csv profiles/${tag}_pmc_traffic.json
  cp gpurun_out/workloads_${tag}.jsonl profiles/${tag}_workloads.jsonl
  cp gpurun_out/bench_${tag}.json profiles/${tag}_bench.json
  for f in chains weights_cold graph_decode attention skinny ln_colsum vocab_gemm; do [ -f gpurun_out/${tag}_$f.log ] && cp gpurun_out/${tag}_$f.log profiles/${tag}_microbench_$f.txt; done
  cp gpurun_out/prof_${tag}r/${tag}r_kernel_stats.csv profiles/${tag}_reconstruct_kernel_stats.csv
  python tools/summ.py profiles/${tag}_reconstruct_kernel_stats.csv 6 40 > profiles/${tag}_reconstruct_kernel_stats_summary.txt
  python tools/native_share.py profiles/${tag}_kernel_stats.csv 6 > profiles/${tag}_native_share.txt
  exit 0
fi
bash tools/prof.sh ${tag}w --steps 4 --warmup 2 || exit 1
bash tools/prof.sh ${tag}l --workload long --steps 2 --warmup 1 || exit 1
bash tools/prof.sh ${tag}r --workload reconstruct --steps 4 --warmup 2 || exit 1
bash tools/pmc.sh FETCH_SIZE ${tag}_fetch || exit 1
bash tools/pmc.sh WRITE_SIZE ${tag}_write || exit 1
bash tools/pmc.sh SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES ${tag}_mfma || exit 1
bash tools/workloads.sh ${tag} reconstruct cycle teacher greedy long > /dev/null || exit 1
timeout -k 10 200 python tools/chains_bench.py 160 2>&1 | grep -v amdgpu > gpurun_out/${tag}_chains.log || exit 1
timeout -k 10 200 python tools/weights_cold_bench.py 160 2>&1 | grep -v amdgpu > gpurun_out/${tag}_weights_cold.log || exit 1
timeout -k 10 200 python tools/graph_decode_bench.py 240 2>&1 | grep -v amdgpu > gpurun_out/${tag}_graph_decode.log || exit 1
timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_attention.log || exit 1
{ timeout -k 10 100 python tools/skinny_bench.py 160 && timeout -k 10 100 python tools/skinny_bench.py 240; } 2>&1 | grep -v amdgpu > gpurun_out/${tag}_skinny.log || exit 1
timeout -k 10 100 python tools/ln_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_ln_colsum.log || exit 1
{ timeout -k 10 100 python tools/vocab_gemm_bench.py; MSSR_PIPE_FIRST_MINM=128 timeout -k 10 100 python tools/vocab_gemm_bench.py; } 2>&1 | grep -v amdgpu > gpurun_out/${tag}_vocab_gemm.log || exit 1
timeout -k 10 300 python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err || exit 1
tail -1 gpurun_out/bench_${tag}.json
