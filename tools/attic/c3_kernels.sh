# per-kernel device times of one C3 run (1 M x 768, cosine, 1024-query batches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3k; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/p -o c -- python bench.py --batch ${2:-1024} --metric ${1:-cosine} --steps 6 --warmup 2 --no-cpu-baseline > $O/log 2>&1
grep -o '"value": [0-9.]*' $O/log | head -1
python tools/rocpd_summary.py $(ls $O/p/*.db $O/p/*/*.db 2>/dev/null | head -1) | grep -E "tsh::" | cut -c1-150
