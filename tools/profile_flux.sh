# rocprofv3 kernel-trace stats of the full-depth FLUX-Kontext edit (tools/bench_flux.py: warm-up edit + timed edit = 16 forwards) -> gpurun_out/<tag>_flux_kernel_stats.csv
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_flux_full -- python3 $R/tools/bench_flux.py > $R/gpurun_out/${TAG}_flux_profiled.log 2>&1
cd $R
f=$(find gpurun_out/trace_flux_full -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_flux_kernel_stats.csv
rm -rf gpurun_out/trace_flux_full
tail -1 gpurun_out/${TAG}_flux_profiled.log | cut -c100-300; head -8 gpurun_out/${TAG}_flux_kernel_stats.csv | cut -c1-170
