#!/bin/bash
# usage (on the GPU box, from the repository root): bash tools/evidence.sh <round tag>
# The bench lines that go with a round's profiles: the default run, two ranks on this one GPU, the pool over one
# and over two contexts, the operators' kernel times, the inflate of every codec, the compressors, large compacted batches whose records change size.  Results under gpurun_out/<tag>_*.
set -e
tag=${1:-r03}
o=gpurun_out
timeout -k 10 400 python3 bench.py > $o/${tag}_default_bench.json 2> $o/${tag}_default_bench.err
echo "default done"
timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu-baseline > $o/${tag}_gpus2.json 2> $o/${tag}_gpus2.err
timeout -k 10 300 python3 bench.py --pool 1 > $o/${tag}_pool1.json 2> $o/${tag}_pool1.err
timeout -k 10 300 python3 bench.py --pool 2 > $o/${tag}_pool2.json 2> $o/${tag}_pool2.err
echo "launcher and pool done"
timeout -k 10 200 python3 tools/storage_time.py > $o/${tag}_storage_time.json 2> $o/${tag}_storage_time.err
timeout -k 10 200 python3 tools/ops_time.py > $o/${tag}_ops_time.json 2> $o/${tag}_ops_time.err
echo "operators done"
for c in snappy lz4 gzip zstd; do
  CODEC=$c timeout -k 10 300 python3 tools/inflate_time.py > $o/${tag}_inflate_$c.json 2> $o/${tag}_inflate_$c.err
done
timeout -k 10 300 python3 tools/deflate_time.py > $o/${tag}_deflate.json 2> $o/${tag}_deflate.err
CODEC=gzip timeout -k 10 300 python3 tools/deflate_time.py > $o/${tag}_deflate_gzip.json 2> $o/${tag}_deflate_gzip.err
CODEC=zstd timeout -k 10 300 python3 tools/deflate_time.py > $o/${tag}_deflate_zstd.json 2> $o/${tag}_deflate_zstd.err
echo "inflate and deflate done"
timeout -k 10 200 python3 tools/gapped_sizes_time.py --batches 32 > $o/${tag}_gapped_sizes_32.json 2> $o/${tag}_gapped_sizes.err
timeout -k 10 200 python3 tools/gapped_sizes_time.py --batches 32 --numbered > $o/${tag}_gapped_sizes_32_numbered.json 2>> $o/${tag}_gapped_sizes.err
timeout -k 10 200 python3 tools/gapped_sizes_time.py > $o/${tag}_gapped_sizes_512.json 2>> $o/${tag}_gapped_sizes.err
timeout -k 10 200 python3 tools/gapped_sizes_time.py --numbered > $o/${tag}_gapped_sizes_512_numbered.json 2>> $o/${tag}_gapped_sizes.err
timeout -k 10 200 python3 tools/gapped_sizes_time.py --short > $o/${tag}_gapped_sizes_512_short.json 2>> $o/${tag}_gapped_sizes.err
timeout -k 10 200 python3 tools/gapped_sizes_time.py --short --numbered > $o/${tag}_gapped_sizes_512_short_numbered.json 2>> $o/${tag}_gapped_sizes.err
echo "compacted batches of changing sizes done"
