cd $GRAFT_REPO_ROOT
bash tools/step_anatomy.sh r04p/new cfg2 > /dev/null 2>&1
export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so LPM_PROJ_DX_STREAM_MIN_N=1024
bash tools/step_anatomy.sh r04p/old cfg2 > /dev/null 2>&1
python - <<'PY'
import re
def load(p):
    rows=[]
    for l in open(p):
        m=re.match(r"\| ([\d.]+) \| ([\d.]+) \| (\d+) \| `(.*)` \|",l)
        if m: rows.append((float(m.group(1)),float(m.group(2)),int(m.group(3)),m.group(4)))
    return rows, open(p).read().splitlines()[2]
for tag in ("new","old"):
    rows,hdr=load(f"gpurun_out/r04p/{tag}_step_cfg2.md")
    print(tag, hdr)
    for key in ("proj_fwd_kernel","proj_reduce","moe_ce_fwd","proj_dx2_kernel","MT256x80x32","fa_update_rows","ca_apply","vlad_bwd_dcentres_k_kernel"):
        for s,d,q,n in rows:
            if key in n: print(f"   {key:28s} start {s:8.1f} dur {d:6.1f} q{q}"); break
    # last kernel end per queue
    for q in (0,1):
        e=max((s+d for s,d,qq,n in rows if qq==q), default=0); print(f"   queue {q} last end {e:.1f}")
PY
