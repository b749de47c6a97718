set -u
ROOTD=$PWD
export NRF_RENDER_LANES=1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  (timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $ROOTD/gpurun_out/ngp_$name -- python3 $ROOTD/bench.py --hash-mode ${MODE:-ngp} --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-isolated 2>&1 | tail -2) > $ROOTD/gpurun_out/ngp_$name.log 2>&1
done
cd $ROOTD
python3 - <<'P'
import csv,glob,collections
for f in glob.glob('gpurun_out/ngp_*/*/*_counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k,r['Counter_Name']]+=1
    for k,v in agg.items():
        if 'hash' in k: print(k, {c:(round(x/cnt[k,c]),cnt[k,c]) for c,x in v.items()})
P
rm -rf gpurun_out/ngp_SQ* 
