cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r04p/new_cfg2 > /dev/null 2>&1
export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so LPM_PROJ_DX_STREAM_MIN_N=1024
bash tools/profile_bench.sh r04p/old_cfg2 > /dev/null 2>&1
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r"\| `(.*?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) ",l)
        if m: d[m.group(1)[:90]]=(int(m.group(2)),float(m.group(3)),float(m.group(4)))
    return d, open(p).read().splitlines()[2]
for mode in ("single","two"):
    a,ha=load(f"gpurun_out/r04p/new_cfg2_{mode}_stream.md"); b,hb=load(f"gpurun_out/r04p/old_cfg2_{mode}_stream.md")
    print(mode, "new:", ha[:110]); print(mode, "old:", hb[:110])
    sa=int(re.search(r"over (\d+) steps",ha).group(1)); sb=int(re.search(r"over (\d+) steps",hb).group(1))
    rows=[]
    for k in set(a)|set(b):
        ta=a.get(k,(0,0,0))[1]/sa*1e3; tb=b.get(k,(0,0,0))[1]/sb*1e3
        rows.append((ta-tb,k,ta,tb))
    rows.sort()
    for d,k,ta,tb in rows[:8]+rows[-8:]:
        print(f"  {d:+8.1f} us/step  new {ta:8.1f} old {tb:8.1f}  {k[:80]}")
PY
