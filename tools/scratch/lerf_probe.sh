set -u
ROOTD=$PWD
export NRF_RENDER_LANES=1
cd /tmp && export TMPDIR=/tmp
(timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/lerfp_stats -- python3 $ROOTD/tools/scratch/lerf_time.py 2>&1 | tail -2) > $ROOTD/gpurun_out/lerfp_stats.log 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  (timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $ROOTD/gpurun_out/lerfp_$name -- python3 $ROOTD/tools/scratch/lerf_time.py 2>&1 | tail -2) > $ROOTD/gpurun_out/lerfp_$name.log 2>&1
done
cd $ROOTD
python3 - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/lerfp_stats/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r['Name'][:80], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
for f in glob.glob('gpurun_out/lerfp_*/*/*_counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-44:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k,r['Counter_Name']]+=1
    for k,v in agg.items():
        if 'lerf' in k or 'hash' in k: print(k, {c:(round(x/cnt[k,c]),cnt[k,c]) for c,x in v.items()})
P
rm -rf gpurun_out/lerfp_stats gpurun_out/lerfp_SQ* gpurun_out/lerfp_GRBM*
