# where the LibTorch-twin (HashEmbedder, fp32 tables) encode spends its time: kernel stats + HBM counters, single lane
set -u
ROOTD=$PWD
export NRF_RENDER_LANES=1
cd /tmp && export TMPDIR=/tmp
A="--hash-mode ngp --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-isolated"
(timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/ngp_stats -- python3 $ROOTD/bench.py $A 2>&1 | tail -2) > $ROOTD/gpurun_out/ngp_stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_')
  (timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $ROOTD/gpurun_out/ngp_$name -- python3 $ROOTD/bench.py --hash-mode ngp --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-isolated 2>&1 | tail -2) > $ROOTD/gpurun_out/ngp_$name.log 2>&1
done
cd $ROOTD
python3 - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/ngp_stats/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r['Name'][:100], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
for g in ('FETCH_SIZE','WRITE_SIZE','TCC_HIT_sum_TCC_MISS_sum'):
    for f in glob.glob('gpurun_out/ngp_%s/*/*_counter_collection.csv'%g):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-60:]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
        for k,v in agg.items():
            if 'hash' in k or 'mlp' in k or 'sigma' in k: print(g, k, {c:(x, cnt[(k,c)]) for c,x in v.items()})
P
rm -rf gpurun_out/ngp_stats gpurun_out/ngp_FETCH_SIZE gpurun_out/ngp_WRITE_SIZE gpurun_out/ngp_TCC_HIT_sum_TCC_MISS_sum
