# the view counters of the detail record's `variants.scenes` beside the search's own ring histogram (option knn_stats), per view
# of a scene: bash tools/r05_cnt.sh [scene] [lines]
cd $GRAFT_REPO_ROOT
PGDVS_KNN_STATS=1 python bench.py --scene ${1:-noisy_depth} --steps 2 --warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --gnt-rays 0 --no-kernel-timing 2> /tmp/err.txt > /dev/null
grep "^bench detail: " /tmp/err.txt | cut -c15- | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:(v['frames_per_s'], v['counters']['knn_queries_to_ring_search'], v['counters']['knn_queries_to_coarse_grid']) for k,v in d['variants']['scenes'].items()})"
grep knn_grid /tmp/err.txt | cut -c1-230 | sort | uniq -c | sort -rn | head -${2:-8}
