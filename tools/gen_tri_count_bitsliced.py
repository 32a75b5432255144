#!/usr/bin/env python3
"""Generate the bit-sliced triangle-count network used by k_face_count.

The number of triangles of a marching-cubes cell (0..5, P3D_TRI_COUNT in tri_table_packed.inc, itself derived from
the reference's case table by tools/gen_tri_table.py) is a function of the 8 corner signs.  With lane = unit the 8
signs of the 64 cells of a unit are 8 machine words, so the three bits of the count can be computed for all cells
of the unit at once by a boolean network: v_bitop3_b32 (any 3-input function, new on gfx950) for the leaves and
the 2:1 selects, common subterms shared.  Which three variables feed the leaves and the order of the five select
variables were found by exhaustive search over all 6720 choices (LEAF_VARS / SELECT_ORDER below: 84 operations; the
natural order x0..x7 needs 100).  The table is not symmetric under complementing all signs, so no variable can be
folded away.
The generator verifies the network against the table for all 256 masks before writing it.
"""
import re
import sys
from pathlib import Path

LEAF_VARS = (0, 1, 6)            # leaves are functions of these corner signs
SELECT_ORDER = (7, 2, 3, 4, 5)    # 2:1 selects, innermost first

ROOT = Path(__file__).resolve().parents[1]
INC = ROOT / "primitive3d_amd" / "csrc" / "tri_table_packed.inc"
OUT = ROOT / "primitive3d_amd" / "csrc" / "tri_count_bitsliced.inc"


def load_counts():
    text = INC.read_text()
    body = text[text.index("#define P3D_TRI_COUNT"):]
    vals = [int(v) for v in re.findall(r"\b\d+\b", body.split("\n", 1)[1])]
    assert len(vals) == 256 and max(vals) == 5 and vals[0] == 0 and vals[255] == 0
    return vals


class Net:
    def __init__(self):
        self.lines, self.cse, self.n = [], {}, 0

    def new(self, expr, key):
        if key in self.cse:
            return self.cse[key]
        name = f"t{self.n}"
        self.n += 1
        self.lines.append(f"    const u32 {name} = {expr};")
        self.cse[key] = name
        return name

    def leaf(self, imm):
        if imm == 0:
            return 0
        if imm == 0xFF:
            return 1
        return self.new(f"P3D_BITOP3(x{LEAF_VARS[2]}, x{LEAF_VARS[1]}, x{LEAF_VARS[0]}, 0x{imm:02x})", ("leaf", imm))

    def mux(self, s, a, b):  # s ? a : b
        if a == b:
            return a
        if a == 1 and b == 0:
            return s
        if a == 0 and b == 1:
            return self.new(f"~{s}", ("not", s))
        if a == 1:
            return self.new(f"{s} | {b}", ("or", s, b))
        if a == 0:
            return self.new(f"~{s} & {b}", ("andn", s, b))
        if b == 0:
            return self.new(f"{s} & {a}", ("and", s, a))
        if b == 1:
            return self.new(f"~{s} | {a}", ("orn", s, a))
        return self.new(f"P3D_BITOP3({s}, {a}, {b}, 0xca)", ("mux", s, a, b))


def evaluate(lines, outs, xs):
    """Interpret the generated straight-line code on python ints (32-bit)."""
    env = {f"x{i}": xs[i] for i in range(8)}
    M = 0xFFFFFFFF

    def bitop3(a, b, c, imm):
        r = 0
        for idx in range(8):
            if (imm >> idx) & 1:
                ta = a if idx & 4 else ~a
                tb = b if idx & 2 else ~b
                tc = c if idx & 1 else ~c
                r |= ta & tb & tc
        return r & M

    for ln in lines:
        m = re.match(r"\s*const u32 (\w+) = (.*);", ln)
        name, expr = m.group(1), m.group(2)
        expr = expr.replace("P3D_BITOP3", "bitop3")
        env[name] = eval(expr, {"bitop3": bitop3}, env) & M
    return [env[o] if isinstance(o, str) else (M if o else 0) for o in outs]


def main() -> int:
    ntri = load_counts()
    def build(cse_muxes):
        net = Net()
        net.cse_muxes = cse_muxes
        outs = []
        for o in range(3):
            def node(level, assign):  # function of the leaf variables and the first `level` select variables
                if level == 0:
                    imm = 0
                    for low in range(8):
                        m = 0
                        for bi, v in enumerate(LEAF_VARS):
                            m |= ((low >> bi) & 1) << v
                        for v, val in assign.items():
                            m |= val << v
                        imm |= ((ntri[m] >> o) & 1) << low
                    return net.leaf(imm)
                v = SELECT_ORDER[level - 1]
                hi = node(level - 1, {**assign, v: 1})   # depth first: few values alive at any time
                lo = node(level - 1, {**assign, v: 0})
                return net.mux(f"x{v}", hi, lo)
            outs.append(node(5, {}))
        return net, outs

    def fix_sel_order(ntri_):
        return ntri_
    net, outs = build(True)
    # exhaustive check: bit position m of the inputs encodes mask m (256 masks = 8 words of 32)
    for base in range(0, 256, 32):
        xs = [sum(((((base + j) >> i) & 1) << j) for j in range(32)) for i in range(8)]
        got = evaluate(net.lines, outs, xs)
        for j in range(32):
            v = sum(((got[o] >> j) & 1) << o for o in range(3))
            assert v == ntri[base + j], (base + j, v, ntri[base + j])
    # registers alive in emission order (depth first: a handful)
    last_use = {}
    for i, ln in enumerate(net.lines):
        for t in re.findall(r"\bt\d+\b", ln.split("=", 1)[1]):
            last_use[t] = i
    for o_ in outs:
        if isinstance(o_, str):
            last_use[o_] = len(net.lines)
    live, max_live = set(), 0
    for i, ln in enumerate(net.lines):
        live.add(re.match(r"\s*const u32 (\w+)", ln).group(1))
        max_live = max(max_live, len(live))
        live = {t for t in live if last_use.get(t, -1) > i}
    lines = []
    for i, ln in enumerate(net.lines):
        lines.append(ln)
    body = "\n".join(lines)
    ops = len(net.lines)
    print("ops", ops, "max live temporaries", max_live)
    text = f"""// GENERATED by tools/gen_tri_count_bitsliced.py -- do not edit.
// Triangle count of 32 cells at once: bit j of x0..x7 = corner sign i of cell j (corner order of the case mask);
// bit j of o0/o1/o2 = bits 0/1/2 of that cell's triangle count.  {ops} word operations, verified against
// P3D_TRI_COUNT for all 256 masks at generation time.
#define P3D_BITOP3(a, b, c, imm) ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), (imm)))
__device__ inline void tri_count_bitsliced(u32 x0, u32 x1, u32 x2, u32 x3, u32 x4, u32 x5, u32 x6, u32 x7, u32& o0,
                                           u32& o1, u32& o2) {{
{body}
    o0 = {outs[0] if isinstance(outs[0], str) else ('~0u' if outs[0] else '0u')};
    o1 = {outs[1] if isinstance(outs[1], str) else ('~0u' if outs[1] else '0u')};
    o2 = {outs[2] if isinstance(outs[2], str) else ('~0u' if outs[2] else '0u')};
}}
#undef P3D_BITOP3
"""
    OUT.write_text(text)
    print("wrote", OUT, "ops:", ops)
    return 0


if __name__ == "__main__":
    sys.exit(main())
