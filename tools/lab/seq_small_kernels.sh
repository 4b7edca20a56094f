cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/seq -o seq --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --layers 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/seq/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-60:] for r in rows]
# run-length encoded sequence of the LAST step (from the last embed_fwd-preceding codes_permute)
starts=[i for i,n in enumerate(names) if 'conv_in_c1' in n]
seq=names[starts[-1]:]
out=[]; prev=None; cnt=0
for nm in seq:
    nm=nm[-48:]
    if nm==prev: cnt+=1
    else:
        if prev is not None: out.append(f"{cnt:3d} x {prev}")
        prev=nm; cnt=1
out.append(f"{cnt:3d} x {prev}")
print(len(seq)); print("\n".join(out))
PY
rm -rf gpurun_out/seq
