"""Scan gfx950 code objects for the "wide VMEM store, then a VALU write of its data registers" hazard.

    python3 tools/store_hazard_scan.py alphasnake-zero_amd/snake_engine/libsnake_engine.so [--json out.json]

The rule (measured on an MI355X by tools/micro/store_hazard.hip, profiles/r6_store_hazard_micro.json):
a VMEM store of more than 64 data bits reads its data registers over several cycles after issue; a VALU instruction that
writes one of them needs WAIT_STATES[kind] wait states between itself and the store.  LLVM inserts them only for a buffer
store WITHOUT an SGPR soffset (GCNHazardRecognizer::createsVALUHazard); the stores WITH one are this repository's to guard
(conv_split.hip's input-gradient epilogue steps from row to row through soffset).  tests/test_store_hazard_cpu.py runs this
over the shipped library: no edit and no compiler may slide an overwriting instruction into the window unseen.

An instruction between the store and the overwrite counts one wait state, `s_nop k` counts k + 1.  The walk follows the
fall-through path and ends at s_endpgm / s_branch / s_setpc (a taken branch refills the instruction buffer: many cycles).
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("SNK_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

# Wait states a VALU write of the data registers needs behind the store, by the store's form.
#   measured (profiles/r6_store_hazard_micro.json): buffer store WITH an SGPR soffset 1 (dwordx3 and x4; x2 has no hazard),
#   buffer store with soffset 0 and global stores (saddr and vaddr) 2 -- wrong data only ever in lanes 8..15 of a group of sixteen.
#   enforced: the SGPR-soffset form, which only this repository's hand-placed s_nop guards, with one wait state of margin (2);
#   the others at what was measured (2), which is exactly what the compiler's own hazard recognizer leaves between them on gfx950
#   (0 of the library's 4 131 such stores have less, 931 have exactly 2: a margin there would mean fighting the compiler);
#   flat and scratch stores were not run by the micro-kernel and are held to the global stores' figure.
MEASURED = {"buffer_sgpr": 1, "buffer_imm": 2, "global": 2}
WAIT_STATES = {"buffer_sgpr": 2, "buffer_imm": 2, "global": 2, "flat": 2, "scratch": 2}

_WIDE = re.compile(r"^(buffer|global|flat|scratch)_store_(dwordx[34]|format_xyzw?|format_d16_xyzw)\b")
_REG = re.compile(r"^([va])(?:(\d+)|\[(\d+):(\d+)\])$")
_END = re.compile(r"^(s_endpgm|s_branch|s_setpc_b64|s_swappc_b64|s_trap)\b")


def _regs(tok):
    """'v12' -> {('v',12)}; 'v[4:7]' -> 4 entries; anything else -> empty"""
    m = _REG.match(tok.strip())
    if not m:
        return set()
    if m.group(2) is not None:
        return {(m.group(1), int(m.group(2)))}
    return {(m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1)}


def _operands(text):
    return [t.strip() for t in text.split(",")] if text else []


def parse_instruction(line):
    """one disassembly line -> (mnemonic, [operands]) or None.  Accepts llvm-objdump -d lines (tab, text, '// addr: enc')."""
    s = line.split("//")[0].strip()
    if not s or s.endswith(":") or s.startswith(("<", ".")) or re.match(r"^[0-9a-f]+ <", s):
        return None
    parts = s.split(None, 1)
    mn = parts[0]
    ops = _operands(parts[1]) if len(parts) > 1 else []
    return mn, ops


def store_info(mn, ops):
    """-> (kind, data register set) for a wide store, else None"""
    m = _WIDE.match(mn)
    if not m:
        return None
    fam = m.group(1)
    if fam == "buffer":
        data = _regs(ops[0])
        soff = ops[3].split()[0] if len(ops) > 3 else "0"
        kind = "buffer_sgpr" if re.match(r"^(s\d+|m0|ttmp\d+)$", soff) else "buffer_imm"
    elif fam == "global":
        data = _regs(ops[1])
        kind = "global"
    elif fam == "flat":
        data = _regs(ops[1])
        kind = "flat"
    else:
        data = _regs(ops[1])
        kind = "scratch"
    return kind, data


def valu_writes(mn, ops):
    """registers a VALU instruction writes (empty for anything that is not VALU or writes only scalars)"""
    if not mn.startswith("v_") or not ops:
        return set()
    w = _regs(ops[0].split()[0])
    if mn.startswith("v_swap_b32") and len(ops) > 1:
        w |= _regs(ops[1].split()[0])
    return w


def wait_states_of(mn, ops):
    if mn == "s_nop":
        try:
            return int(ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def scan_text(text, wait_states=None, source="<text>"):
    """-> dict(stores=..., by_kind=..., violations=[...], slid=[...]).  `slid`: SGPR-soffset buffer stores whose FIRST following
    instruction is a VALU instruction (not an overwrite) -- legal, reported because the hand-placed s_nop was meant to sit there."""
    ws_need = dict(WAIT_STATES)
    if wait_states:
        ws_need.update(wait_states)
    lines = text.splitlines()
    func = "?"
    insts = []
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:\s*$", ln)
        if m:
            func = m.group(1)
            insts.append(None)               # a function boundary ends every walk
            continue
        p = parse_instruction(ln)
        if p:
            insts.append((p[0], p[1], func, ln.split("//")[0].strip()))
    out = {"source": source, "stores": 0, "by_kind": {}, "violations": [], "slid": [], "guarded_by_nop": 0}
    horizon = max(ws_need.values())
    for i, it in enumerate(insts):
        if it is None:
            continue
        si = store_info(it[0], it[1])
        if not si:
            continue
        kind, data = si
        out["stores"] += 1
        out["by_kind"][kind] = out["by_kind"].get(kind, 0) + 1
        need = ws_need[kind]
        ws, j, between = 0, i + 1, []
        if kind == "buffer_sgpr" and j < len(insts) and insts[j] is not None:
            if insts[j][0] == "s_nop":
                out["guarded_by_nop"] += 1
            elif insts[j][0].startswith("v_"):
                out["slid"].append({"function": it[2], "store": it[3], "next": insts[j][3]})
        while j < len(insts) and insts[j] is not None and ws < max(need, horizon):
            mn, ops, _, txt = insts[j]
            if _END.match(mn):
                break
            hit = valu_writes(mn, ops) & data
            if hit and ws < need:
                out["violations"].append({"function": it[2], "kind": kind, "store": it[3], "overwrite": txt, "wait_states": ws,
                                          "needed": need, "between": between[:]})
                break
            between.append(txt)
            ws += wait_states_of(mn, ops)
            j += 1
    return out


def code_objects(path, workdir):
    """the gfx950 code objects inside a host .so / .o (llvm-objdump --offloading writes them next to its input: work on a copy)"""
    local = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if f.startswith(os.path.basename(path) + ".") and "gfx950" in f)


def disassemble(obj):
    return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", obj], check=True, stdout=subprocess.PIPE, text=True).stdout


def scan_library(path, wait_states=None):
    total = {"library": os.path.basename(path), "code_objects": 0, "stores": 0, "by_kind": {}, "violations": [], "slid": [], "guarded_by_nop": 0,
             "wait_states_required": dict(WAIT_STATES, **(wait_states or {}))}
    with tempfile.TemporaryDirectory() as d:
        objs = code_objects(path, d)
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {path}")
        for o in objs:
            r = scan_text(disassemble(o), wait_states, os.path.basename(o))
            total["code_objects"] += 1
            total["stores"] += r["stores"]
            total["guarded_by_nop"] += r["guarded_by_nop"]
            for k, v in r["by_kind"].items():
                total["by_kind"][k] = total["by_kind"].get(k, 0) + v
            total["violations"] += r["violations"]
            total["slid"] += r["slid"]
    return total


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    res = scan_library(sys.argv[1])
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(res, f, indent=1)
    brief = {k: v for k, v in res.items() if k not in ("violations", "slid")}
    brief["violations"] = len(res["violations"])
    brief["slid"] = len(res["slid"])
    print(json.dumps(brief))
    for v in res["violations"][:20]:
        print("VIOLATION", json.dumps(v))
    return 1 if res["violations"] else 0


if __name__ == "__main__":
    sys.exit(main())
