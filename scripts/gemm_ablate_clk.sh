#!/bin/bash
# cycles (GRBM_GUI_ACTIVE) and wall time per launch for the full kernel and each ablation library -> effective clock
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for n in 0 1 2 3 5 6; do
  lib=$R/touchnet_amd/_lib/variants/abl$n/libtouchnet_amd.so; [ $n = 0 ] && lib=$R/touchnet_amd/_lib/libtouchnet_amd.so
  TN_AMD_LIB=$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/gemm_clk_r03/abl$n --output-format csv -- python $R/scripts/gemm_prof.py 3 > $R/gpurun_out/gemm_clk_abl$n.log 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for n in (0,1,2,3,5,6):
    rows=collections.defaultdict(dict)
    for p in glob.glob(f'gpurun_out/gemm_clk_r03/abl{n}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            k=(r['Kernel_Name'][:48], r['Dispatch_Id'])
            rows[k][r['Counter_Name']]=float(r['Counter_Value'])
            rows[k]['us']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg=collections.defaultdict(list)
    for (name,_),v in rows.items():
        if 'gemm_kernel' in name or 'Cijk' in name: agg[name].append(v)
    print(f"== ablation {n}")
    for name,l in agg.items():
        l.sort(key=lambda v:v['us'])
        for grp in (l[:3], l[3:6]) if len(l)>=6 else (l,):
            if not grp: continue
            m=lambda c: sum(v.get(c,0) for v in grp)/len(grp)
            print(f"  {name:48s} {m('us'):7.0f} us  GRBM {m('GRBM_GUI_ACTIVE'):.3g}  clk {m('GRBM_GUI_ACTIVE')/8/m('us')/1e3:.2f} GHz  parked {m('SQ_WAIT_ANY')/max(1,m('SQ_WAVE_CYCLES')):.2f}  issue-stall {m('SQ_WAIT_INST_ANY')/max(1,m('SQ_WAVE_CYCLES')):.2f}  active {m('SQ_ACTIVE_INST_ANY')/max(1,m('SQ_WAVE_CYCLES')):.2f}")
PY
