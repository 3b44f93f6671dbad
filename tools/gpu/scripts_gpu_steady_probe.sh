cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/steady_prof
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/steady_probe.py > $out/probe.txt 2> $out/rocprof.err
echo "rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
cat $out/probe.txt
python3 tools/kstats.py $out/kernel_stats.csv
