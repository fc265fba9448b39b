#!/bin/bash
# Round profile set.  On the GPU box from the repo root, one gpurun call per stage (a call is limited to 20 minutes):
#   bash tools/collect_profiles.sh r03 stage1   kernel stats of the headline step / --workload long / reconstruct, the
#                                               FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes of the headline step
#   bash tools/collect_profiles.sh r03 stage2   FETCH_SIZE / WRITE_SIZE passes of --workload long
#   (container)  bash tools/collect_profiles.sh r03 traffic   -> profiles/r03_pmc_traffic.json, which stage3's bench lines read
#   bash tools/collect_profiles.sh r03 stage3   every SURVEY 8d workload line, microbenchmarks, host time, phase split,
#                                               the critics' phase kernel list, the full bench line with cpu_baseline
#   (container)  bash tools/collect_profiles.sh r03 copy      summaries -> profiles/
# Every PMC stage records the hash of the sources it ran on (tools/pmc_traffic.py source_sha16); do not edit the package
# between stage1 and stage3.
tag=$1; stage=$2
sha() { python -c "import sys; sys.path.insert(0, 'tools'); from pmc_traffic import source_sha16; print(source_sha16('.'))"; }
case $stage in
stage1)
  bash tools/prof.sh ${tag}w --steps 4 --warmup 2 || exit 1
  bash tools/prof.sh ${tag}l --workload long --steps 2 --warmup 1 || exit 1
  bash tools/prof.sh ${tag}r --workload reconstruct --steps 4 --warmup 2 || exit 1
  bash tools/pmc.sh FETCH_SIZE ${tag}_fetch || exit 1
  bash tools/pmc.sh WRITE_SIZE ${tag}_write || exit 1
  bash tools/pmc.sh SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES ${tag}_mfma || exit 1
  sha > gpurun_out/${tag}_sha_wgan.txt
  ;;
stage2)
  # sweeps launched eagerly: counter collection over this workload's replayed graphs (140-step sweeps) crashed rocprofv3
  MSSR_DECODE_GRAPH=0 bash tools/pmc.sh FETCH_SIZE ${tag}_long_fetch --workload long || exit 1
  MSSR_DECODE_GRAPH=0 bash tools/pmc.sh WRITE_SIZE ${tag}_long_write --workload long || exit 1
  sha > gpurun_out/${tag}_sha_long.txt
  ;;
traffic)
  rm -f profiles/${tag}_pmc_traffic.json
  cp gpurun_out/pmc_${tag}_fetch.csv profiles/${tag}_pmc_fetch_size.csv
  cp gpurun_out/pmc_${tag}_write.csv profiles/${tag}_pmc_write_size.csv
  cp gpurun_out/pmc_${tag}_mfma.csv profiles/${tag}_pmc_mfma_busy.csv
  python tools/pmc_traffic.py wgan profiles/${tag}_pmc_fetch_size.csv profiles/${tag}_pmc_write_size.csv profiles/${tag}_pmc_traffic.json $(cat gpurun_out/${tag}_sha_wgan.txt) || exit 1
  if [ -f gpurun_out/pmc_${tag}_long_fetch.csv ]; then
    cp gpurun_out/pmc_${tag}_long_fetch.csv profiles/${tag}_long_pmc_fetch_size.csv
    cp gpurun_out/pmc_${tag}_long_write.csv profiles/${tag}_long_pmc_write_size.csv
    python tools/pmc_traffic.py long profiles/${tag}_long_pmc_fetch_size.csv profiles/${tag}_long_pmc_write_size.csv profiles/${tag}_pmc_traffic.json $(cat gpurun_out/${tag}_sha_long.txt) || exit 1
  fi
  ;;
stage3)
  bash tools/workloads.sh ${tag} reconstruct cycle teacher greedy long > /dev/null || exit 1
  timeout -k 10 200 python tools/chains_bench.py 160 2>&1 | grep -v amdgpu > gpurun_out/${tag}_chains.log || exit 1
  timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_attention.log || exit 1
  { timeout -k 10 100 python tools/skinny_bench.py 160 && timeout -k 10 100 python tools/skinny_bench.py 240; } 2>&1 | grep -v amdgpu > gpurun_out/${tag}_skinny.log || exit 1
  { timeout -k 10 100 python tools/skinny_bench.py 240; MSSR_SKINNY_DMA=0 timeout -k 10 100 python tools/skinny_bench.py 240; } 2>&1 | grep -v amdgpu > gpurun_out/${tag}_skinny_dma_on_off.log || exit 1
  timeout -k 10 100 python tools/pipe_mid_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_pipe_mid.log || exit 1
  timeout -k 10 300 python tools/host_time.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_host_time.txt || exit 1
  timeout -k 10 200 python tools/phase_wall.py 8 2>&1 | grep -v amdgpu > gpurun_out/${tag}_phase_wall.txt || exit 1
  timeout -k 10 200 python tools/sweep_bench.py 5 2>&1 | grep -v amdgpu > gpurun_out/${tag}_sweep.txt || exit 1
  timeout -k 10 200 python tools/graph_host_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_graph_host.txt || exit 1
  bash tools/dphase_prof.sh ${tag} || exit 1
  timeout -k 10 400 python tools/dp_overhead.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_dp_single_rank.txt || exit 1
  timeout -k 10 300 python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err || exit 1
  tail -1 gpurun_out/bench_${tag}.json
  ;;
copy)
  cp gpurun_out/prof_${tag}w/${tag}w_kernel_stats.csv profiles/${tag}_kernel_stats.csv
  cp gpurun_out/prof_${tag}l/${tag}l_kernel_stats.csv profiles/${tag}_long_kernel_stats.csv
  cp gpurun_out/prof_${tag}r/${tag}r_kernel_stats.csv profiles/${tag}_reconstruct_kernel_stats.csv
  python tools/summ.py profiles/${tag}_kernel_stats.csv 6 60 > profiles/${tag}_kernel_stats_summary.txt
  python tools/summ.py profiles/${tag}_long_kernel_stats.csv 3 40 > profiles/${tag}_long_kernel_stats_summary.txt
  python tools/summ.py profiles/${tag}_reconstruct_kernel_stats.csv 6 40 > profiles/${tag}_reconstruct_kernel_stats_summary.txt
  python tools/native_share.py profiles/${tag}_kernel_stats.csv 6 > profiles/${tag}_native_share.txt
  cp gpurun_out/workloads_${tag}.jsonl profiles/${tag}_workloads.jsonl
  cp gpurun_out/bench_${tag}.json profiles/${tag}_bench.json
  for f in chains attention skinny skinny_dma_on_off pipe_mid; do [ -f gpurun_out/${tag}_$f.log ] && cp gpurun_out/${tag}_$f.log profiles/${tag}_microbench_$f.txt; done
  for f in host_time phase_wall sweep graph_host dp_single_rank; do [ -f gpurun_out/${tag}_$f.txt ] && cp gpurun_out/${tag}_$f.txt profiles/${tag}_$f.txt; done
  [ -f gpurun_out/dphase_${tag}.txt ] && cp gpurun_out/dphase_${tag}.txt profiles/${tag}_dphase_kernels.txt
  ;;
*) echo "usage: collect_profiles.sh <tag> stage1|stage2|traffic|stage3|copy"; exit 2;;
esac
