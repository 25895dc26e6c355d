"""Looks through a gfx950 assembly listing (hipcc -S --cuda-device-only) for pieces of a spilled
register tuple that are saved and never restored.

When the register allocator evicts a multi-register value it may save the pieces in different
places: some to scratch memory ("Folded Spill"), some to a free register with a copy marked
"Reload Reuse".  The one-step LSTM layer kernel with its lower-layer weights pinned to
accumulation registers compiles to such a split (one 128-bit weight vector: two dwords to scratch,
two to v254 / v255), and the code that rebuilds the vector before the matrix instruction restores
three of the four dwords only: the register holding the fourth is never read, and the
instruction multiplies by whatever its operand register held before.  The value computed is
wrong or right depending on that stale content, which is why the failure came and went with the
inputs and with every change of the surrounding code.

usage: orphan_spills.py listing.s [--any]     exit status 1 when something is found
  --any: report every spill or reload as a finding (for objects that must have none)."""
import re
import sys


def registers(text):
    """Indices of the ordinary vector registers named in an operand string."""
    found = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        found.update(range(int(a), int(b) + 1))
    found.update(int(n) for n in re.findall(r"\bv(\d+)\b", text))
    return found


def kernels(lines):
    name, body = None, []
    for no, line in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append((no, line))
            if line.strip().startswith("s_endpgm") or line.startswith(".Lfunc_end"):
                if body:
                    yield name, body
                name, body = None, []


def reads_of(body):
    """Map register index -> line numbers where an instruction reads it."""
    reads = {}
    for no, line in body:
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        parts = code.split(None, 1)
        if len(parts) < 2:
            continue
        op, operands = parts
        ops = [o.strip() for o in operands.split(",")]
        stores = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store"))
        sources = ops if stores else ops[1:]
        for r in registers(" ".join(sources)):
            reads.setdefault(r, []).append(no)
    return reads


def main():
    path = sys.argv[1]
    report_any = "--any" in sys.argv[2:]
    lines = open(path).read().splitlines()
    findings = 0
    for name, body in kernels(lines):
        reads = reads_of(body)
        spills = [(no, l) for no, l in body if "Spill" in l or "Reload" in l]
        if report_any and spills:
            findings += len(spills)
            print(f"{name}: {len(spills)} spill / reload instructions, first at line {spills[0][0]}")
        for no, line in body:
            m = re.match(r"\s*v_accvgpr_read_b32 v(\d+), a(\d+)\s*;\s*Reload Reuse", line)
            if not m:
                continue
            saved, src = int(m.group(1)), int(m.group(2))
            later = [r for r in reads.get(saved, []) if r > no]
            if not later:
                findings += 1
                print(f"{name}: line {no}: a{src} saved to v{saved}, and v{saved} is never read afterwards")
    print(f"{path}: {findings} finding(s)")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
