#!/usr/bin/env python
"""Scan the gfx950 ISA of libdram_hip.so for the store hazard class met in round 3 ("two hipcc 7.2 faults"):

    buffer_store_dwordx4 v[104:107], ..., s58 offen      ; store data = v104..v107, soffset in an SGPR
    v_mov_b32 v104, v4                                   ; the NEXT instruction overwrites a data register

The CDNA ISA rule behind it ("VMEM store more than 64 bits of data, followed by a write of the store's data VGPRs: 1 wait
state required") is enforced by LLVM's hazard recogniser only when the store's `soffset` is NOT an SGPR (for an SGPR soffset
the hardware reads the data in a later cycle and the recogniser assumes that is early enough) -- measured on gfx950 it is not:
conv3d_k3_fwd_c1w_kernel wrote the next channel's value into ~0.7 % of its first dwords at 128^3.  The kernel now pins two
wait states behind each such store; this script fails when ANY kernel of the library holds a > 64-bit buffer / global / flat
store (dwordx3 / dwordx4) whose data registers are written by a VALU instruction with fewer than one wait state in between
(s_nop N counts N + 1; any other instruction counts one).

    python scripts/isa_hazards.py [path/to/libdram_hip.so]        exit status 1 when a hazard is found

Pure host tool: objcopy + clang-offload-bundler + llvm-objdump from /opt/rocm (no GPU).  tests/test_host_cpu.py runs it."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
VREG_RANGE = re.compile(r"v\[(\d+):(\d+)\]")
VREG = re.compile(r"\bv(\d+)\b")
SNOP = re.compile(r"^\s*s_nop\s+(\d+)")


def code_objects(lib, tmp):
    """Every gfx950 code object bundled in `lib` (one per translation unit), as files."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(blob[s:e])
        co = os.path.join(tmp, f"code{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={part}",
                            f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def data_registers(store_line):
    """VGPR numbers of the store's DATA operand (the first operand of buffer stores, the second of global / flat ones)."""
    m = STORE.match(store_line)
    op, rest = m.group(1), m.group(2)
    ops = [o.strip() for o in rest.split(",")]
    data = ops[0] if op.startswith("buffer_") or op.startswith("scratch_") and False else None
    if op.startswith("buffer_"):
        data = ops[0]
    else:                               # global_store / flat_store / scratch_store: vaddr, vdata, ...
        data = ops[1] if len(ops) > 1 else ops[0]
    r = VREG_RANGE.search(data)
    if r:
        return set(range(int(r.group(1)), int(r.group(2)) + 1))
    v = VREG.search(data)
    return {int(v.group(1))} if v else set()


def sgpr_soffset(store_line):
    """True when a buffer store's soffset operand is an SGPR (the case LLVM's recogniser skips)."""
    m = STORE.match(store_line)
    if not m.group(1).startswith("buffer_"):
        return False
    ops = [o.strip() for o in m.group(2).split(",")]
    # buffer_store vdata, vaddr|off, srsrc, soffset [modifiers]
    if len(ops) < 4:
        return False
    so = ops[3].split()[0]
    return bool(re.match(r"^s\d+$", so)) or so in ("m0",) or so.startswith("ttmp")


def written_registers(line):
    """VGPRs a VALU instruction writes (its first operand; both operands of the swaps)."""
    t = line.strip().split(None, 1)
    if len(t) < 2:
        return set()
    op, rest = t
    if op.startswith(("s_", "buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store", "v_cmp", "v_nop")):
        return set()
    # VALU only: the ISA's rule is about a VALU write in the cycle after the store's issue.  A load (VMEM, LDS) that targets
    # the store's data registers writes them a memory latency (>= tens of cycles) later -- the store has long read its data
    # (hipcc emits that pair freely, e.g. in the slab epilogue of conv3d_k3_wgrad_wzy_kernel, bit-exact in every test).
    if not op.startswith("v_"):
        return set()
    first = rest.split(",")[0].strip()
    r = VREG_RANGE.match(first)
    if r:
        return set(range(int(r.group(1)), int(r.group(2)) + 1))
    v = re.match(r"^v(\d+)$", first)
    out = {int(v.group(1))} if v else set()
    if op.startswith("v_permlane") or op == "v_swap_b32":        # both operands are written
        second = rest.split(",")[1].strip() if "," in rest else ""
        v2 = re.match(r"^v(\d+)$", second)
        if v2:
            out.add(int(v2.group(1)))
    return out


def scan(co):
    asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    kernel = "?"
    lines = []
    for raw in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw)
        if m:
            kernel = m.group(1)
            continue
        body = raw.split("//")[0].rstrip()
        if body.strip():
            lines.append((kernel, body))
    hazards, stores, sgpr_stores = [], 0, 0
    for i, (kern, body) in enumerate(lines):
        if not STORE.match(body):
            continue
        stores += 1
        sg = sgpr_soffset(body)
        sgpr_stores += sg
        data = data_registers(body)
        waits = 0
        for kern2, nxt in lines[i + 1:i + 4]:
            if kern2 != kern or waits >= 1:
                break
            n = SNOP.match(nxt)
            if n:
                waits += int(n.group(1)) + 1
                continue
            hit = written_registers(nxt) & data
            if hit:
                hazards.append((kern, body.strip(), nxt.strip(), sorted(hit), sg))
                break
            waits += 1                  # any other instruction is one wait state
    return stores, sgpr_stores, hazards


def main(lib):
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(lib, tmp)
        if not cos:
            print(f"isa_hazards: no {TARGET} code object found in {lib}", file=sys.stderr)
            return 2
        total = total_sg = 0
        found = []
        for co in cos:
            st, sg, hz = scan(co)
            total += st
            total_sg += sg
            found += hz
    print(f"isa_hazards: {len(cos)} code objects, {total} stores of more than 64 bits ({total_sg} with an SGPR soffset), "
          f"{len(found)} followed at once by a write of their data registers")
    for kern, st, nxt, regs, sg in found:
        print(f"  {kern}\n      {st}\n      {nxt}      <- writes v{regs}{'  (SGPR soffset: not separated by the compiler)' if sg else ''}")
    return 1 if found else 0


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "bodyct-dram_amd", "libdram_hip.so")))
