# true kernel durations (rocprofv3 kernel trace) of the GEMM kernels of kbench pw_big under diagnosis flags
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 1" "13 1" "0 0" "7 0"; do
  set -- $cfg
  O=$R/gpurun_out/r2g/prof_d$1_s$2; rm -rf $O; mkdir -p $O
  PB_GEMM_DMA_DBG=$1 PB_GEMM_STREAM=$2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o kb -- python3 $R/tools/kbench.py pw_big > $O/run.log 2>&1 || { echo failed $cfg; tail -5 $O/run.log; exit 1; }
  echo "== DBG=$1 STREAM=$2"
  python3 - "$O" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'pw_' not in n: continue
    key=(n.split('(')[0][-60:], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    v.sort()
    print('%-64s grid %-8s n=%3d median %8.1f us min %8.1f'%(k[0],k[1],len(v),v[len(v)//2],v[0]))
PY
done
