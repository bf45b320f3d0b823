for sg in 16 32; do for b in 15625 31250 62500 125000 250000; do
MN_LEIDEN_SG=$sg MN_LEIDEN_BATCH=$b python bench_graph.py --workload leiden --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('sg',$sg,'batch',$b,'ms',round(j['ms_per_step'],2),'Q',round(j['modularity'],5),'sweeps',j['sweeps'],'nmi',round(j['nmi_vs_planted'],4))"
done; done
