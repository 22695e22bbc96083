# The rocprofv3 passes behind profiles/r02_*: run on a GPU box from the repository root
# (gpurun -- bash tools/profile_passes.sh); then python3 tools/condense_profiles.py gpurun_out/r02final writes the summaries under profiles/.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final
mkdir -p $O
export TMPDIR=/tmp
# the profiling build of the library for the phase-by-phase figures of plan_cells_kernel
make -C azplugins_amd/csrc variant SRC=pair_plan_cells NAME=pcprof DEFS=-DAZP_PLAN_CELLS_PROFILE > $O/pcprof_build.log 2>&1
B="python3 bench.py --no-cpu-baseline --no-side-figures --steps 80 --warmup 8"
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- $B > $O/bench_stats.json 2> $O/bench_stats.err
echo "bench stats pass done"
rocprofv3 --kernel-trace --stats -d $O/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > $O/md_bench.log 2> $O/md_bench.err
tail -3 $O/md_bench.log
python3 tools/md_bench.py --steps 300 > $O/md_bench_noprof.log 2>&1
tail -3 $O/md_bench_noprof.log
for s in 0 1 2 514 4; do AZP_LIB_PATH=tools/libazp_pcprof.so AZP_PLAN_CELLS_STOP=$s python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-60; done > $O/plan_cells_phases.log
python3 tools/plan_cells_probe.py --melt 100 2>&1 | tail -1 | cut -c1-60 >> $O/plan_cells_phases.log
cat $O/plan_cells_phases.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/pc_sq --output-format csv -- python3 tools/plan_cells_probe.py --reps 3 > $O/pc_sq.log 2> $O/pc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pc_fetch --output-format csv -- python3 tools/plan_cells_probe.py --reps 3 > $O/pc_fetch.log 2> $O/pc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pc_write --output-format csv -- python3 tools/plan_cells_probe.py --reps 3 > $O/pc_write.log 2> $O/pc_write.err
rocprofv3 --kernel-trace --stats -d $O/c4_stats --output-format csv -- python3 tools/xtiled_probe.py c4 > $O/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5_stats --output-format csv -- python3 tools/xtiled_probe.py c5 > $O/c5.log 2>&1
tail -1 $O/c4.log | cut -c1-300; tail -1 $O/c5.log | cut -c1-300
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
python3 tools/summarize_prof.py $O $O/all > /dev/null 2>&1 || true
echo done
