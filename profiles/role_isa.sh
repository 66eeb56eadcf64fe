#!/bin/bash
# Static instruction mix of every role of the five-wave fused kernel (no GPU needed): dn_kernels_mw.hip compiled to assembly with
# -DDN_ROLE_MARK (a comment at the top of every role's K-step loop); the instructions of each role's loop of
# dn_step_many_5w_kernel<double, false> -- from its mark to the backward branch that closes it -- are counted by class.  Rarely taken
# blocks (resets, gimbal lock, clamps) are inside the count: it is the code's size by role, not the dynamic mix (that is instmix.sh).
#   profiles/role_isa.sh [extra hipcc flags]  ->  stdout
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math --cuda-device-only -S \
      -DDN_ROLE_MARK "$@" $ROOT/drl-dronenavigation_amd/csrc/dn_kernels_mw.hip -o $T/mw.s 2>/dev/null
python3 - $T/mw.s <<'PY'
import re, sys
from collections import Counter
kern = "_ZN12_GLOBAL__N_122dn_step_many_5w_kernelIdLb0EEEv8DnParams8DnStepIOi"
body, on = [], False
for l in open(sys.argv[1]):
    if l.startswith(kern + ":"): on = True; continue
    if on:
        if l.startswith(".Lfunc_end"): break
        body.append(l.rstrip("\n"))
labels, ins, marks = {}, [], {}
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = len(ins); continue
    t = l.strip()
    m = re.match(r"; DN_ROLE_LOOP (\w)", t)
    if m: marks[m.group(1)] = len(ins); continue
    if l.startswith("\t") and t and not t.startswith((".", ";")): ins.append(t)
print(f"kernel: {len(ins)} instructions;  per role, the K-step loop:")
print(f"{'role':4} {'loop':>5} | {'valu':>5} {'f64':>4} {'cvt':>4} {'f32':>4} {'int':>4} {'mov':>4} {'cnd':>4} {'cmp':>4} {'trans':>5} | {'salu':>5} {'lds':>4} {'vmem':>4} {'scratch':>7} {'branch':>6}")
tot = Counter()
for name in "LAQNX":
    start = marks[name]
    # the loop's header is the last label at or before the mark; its closing branch is the LAST backward branch to a label <= start
    hdr = max(v for v in labels.values() if v <= start)
    end = None
    for i in range(start, len(ins)):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ins[i])
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= start and tgt >= hdr - 400: end = i
        if i > start and any(i == v for k, v in marks.items() if k != name): break
    loop = ins[hdr:end + 1]
    c = Counter(x.split()[0] for x in loop)
    def cnt(pred): return sum(v for k, v in c.items() if pred(k))
    tr = ("v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
    valu = cnt(lambda k: k.startswith("v_"))
    f64 = cnt(lambda k: k.startswith("v_") and "f64" in k and not k.startswith(("v_cvt", "v_cmp") + tr))
    cvt = cnt(lambda k: k.startswith("v_cvt"))
    trans = cnt(lambda k: k.startswith(tr))
    mov = cnt(lambda k: k.startswith(("v_mov", "v_accvgpr")))
    cnd = cnt(lambda k: k.startswith("v_cndmask"))
    cmp_ = cnt(lambda k: k.startswith("v_cmp"))
    f32 = cnt(lambda k: k.startswith("v_") and "f32" in k and not k.startswith(("v_cvt", "v_cmp") + tr))
    other = valu - f64 - cvt - trans - mov - cnd - cmp_ - f32
    row = dict(loop=len(loop), valu=valu, f64=f64, cvt=cvt, f32=f32, int=other, mov=mov, cnd=cnd, cmp=cmp_, trans=trans,
               salu=cnt(lambda k: k.startswith("s_") and not k.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop"))),
               lds=cnt(lambda k: k.startswith("ds_")), vmem=cnt(lambda k: k.startswith(("global_", "buffer_"))),
               scratch=cnt(lambda k: k.startswith("scratch_")), branch=cnt(lambda k: k.startswith(("s_cbranch", "s_branch"))))
    tot.update(row)
    print(f"{name:4} {row['loop']:5d} | {valu:5d} {f64:4d} {cvt:4d} {f32:4d} {other:4d} {mov:4d} {cnd:4d} {cmp_:4d} {trans:5d} | {row['salu']:5d} {row['lds']:4d} {row['vmem']:4d} {row['scratch']:7d} {row['branch']:6d}")
print(f"{'sum':4} {tot['loop']:5d} | {tot['valu']:5d} {tot['f64']:4d} {tot['cvt']:4d} {tot['f32']:4d} {tot['int']:4d} {tot['mov']:4d} {tot['cnd']:4d} {tot['cmp']:4d} {tot['trans']:5d} | {tot['salu']:5d} {tot['lds']:4d} {tot['vmem']:4d} {tot['scratch']:7d} {tot['branch']:6d}")
PY
rm -rf $T
