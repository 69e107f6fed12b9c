#!/bin/bash
# cycles (GRBM_GUI_ACTIVE), wall time and wave-state shares per launch for a list of variant libraries
# usage: scripts/gemm_clk.sh name1 name2 ...   (names under touchnet_amd/_lib/variants/, "product" = the product library)
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  lib=$R/touchnet_amd/_lib/variants/$n/libtouchnet_amd.so; [ $n = product ] && lib=$R/touchnet_amd/_lib/libtouchnet_amd.so
  TN_AMD_LIB=$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/gemm_clk/$n --output-format csv -- python $R/scripts/gemm_prof.py 3 > $R/gpurun_out/gemm_clk_$n.log 2>&1
done
cd $R
python3 - "$@" <<'PY'
import csv,glob,collections,sys
for n in sys.argv[1:]:
    rows=collections.defaultdict(dict)
    for p in glob.glob(f'gpurun_out/gemm_clk/{n}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            k=(r['Kernel_Name'][:48], r['Dispatch_Id'])
            rows[k][r['Counter_Name']]=float(r['Counter_Value'])
            rows[k]['us']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg=collections.defaultdict(list)
    for (name,_),v in rows.items():
        if 'gemm' in name or 'Cijk' in name: agg[name].append(v)
    print(f"== {n}")
    for name,l in agg.items():
        l.sort(key=lambda v:v['us'])
        for grp in (l[:3], l[3:6]) if len(l)>=6 else (l,):
            if not grp: continue
            m=lambda c: sum(v.get(c,0) for v in grp)/len(grp)
            print(f"  {name:48s} {m('us'):7.0f} us  GRBM {m('GRBM_GUI_ACTIVE'):.3g}  clk {m('GRBM_GUI_ACTIVE')/8/m('us')/1e3:.2f} GHz  parked {m('SQ_WAIT_ANY')/max(1,m('SQ_WAVE_CYCLES')):.2f}  issue-stall {m('SQ_WAIT_INST_ANY')/max(1,m('SQ_WAVE_CYCLES')):.2f}  active {m('SQ_ACTIVE_INST_ANY')/max(1,m('SQ_WAVE_CYCLES')):.2f}  lds_inst {m('SQ_INSTS_LDS'):.3g} conflicts {m('SQ_LDS_BANK_CONFLICT'):.3g}")
PY
