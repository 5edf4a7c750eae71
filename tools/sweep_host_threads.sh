for t in ${THREADS:-16 14 15 17 20 24 16}; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --placement-trials 0 --host-threads $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']
print('threads $t e2e %.4g wall %.3f entropy_share %.3f'%(e['value'],e['wall_s'],e['stages_rank0']['entropy_decode_host']['share_of_wall']))"
done
