# Copy the judged summaries of a tools/refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/ (tracked).
# usage: bash tools/collect_profiles.sh r02
tag=${1:?tag}
src=gpurun_out/$tag
cp $src/bench.json profiles/${tag}_bench.json
cp $src/configs.jsonl profiles/${tag}_configs.jsonl
cp $src/prof/bench_kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp $src/profc/configs_kernel_stats.csv profiles/${tag}_configs_kernel_stats.csv
python3 tools/traffic_json.py $src/pmcf/p_counter_collection.csv $src/pmcw/p_counter_collection.csv profiles/traffic.json > profiles/${tag}_pmc_fetch_write.txt
cp profiles/traffic.json profiles/${tag}_traffic.json
tail -1 $src/gpu_tests.log > profiles/${tag}_gpu_tests.txt
