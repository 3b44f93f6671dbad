#!/bin/bash
# rehearsal of the N > 1 bench path on the one-GPU box: 2 ranks over gloo sharing cuda:0, both scaling modes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r02_2rank}
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
rc=$?; tail -3 $out/pytest_gpu.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit $rc
for mode in weak strong; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --backend gloo --share-device --scaling $mode --no-cpu-baseline --no-extras > $out/bench_2rank_$mode.json 2> $out/bench_2rank_$mode.err
  echo "$mode rc=$?"
  python3 -c "
import json; d=json.loads(open('$out/bench_2rank_$mode.json').read().strip().splitlines()[-1])
print('$mode', 'value %.4g ms/step %.4f batch/gpu %s' % (d['value'], d['ms_per_step'], d['config']['batch_per_gpu']), 'other:', {k: d['other_scaling'][k] for k in ('scaling','value','ms_per_step','batch_per_gpu')})
"
done
