#!/usr/bin/env python3
"""Instruction audit of the bidiagonalisation loops in a `hipcc -S` listing of csrc/car.hip (round 6, review item 2).

    hipcc ... -S --cuda-device-only -o car.s car.hip ;  bidiag_isa_audit.py car.s [kernel-substring]

For every top-level loop of the kernel that holds two s_barrier (one step of car_bidiag2_block<S>) it prints the
instructions of one iteration by class next to the count the SOURCE asks for (4 nk nq matrix FMAs, the vector work, the
row sums' DPP moves, the LDS traffic), so that what no source line shows stands out: register moves, AGPR round trips,
waterfall loops, s_nop padding, selects and compares."""
import collections
import re
import sys

CLASSES = [
    ("fp64 fma/mul/add", r"v_(fma|mul|add|fmac|max|min|rsq|rcp|sqrt|ldexp|frexp|div_\w+|trig)\w*_f64"),
    ("dpp mov", r"v_mov_b32_dpp"),
    ("v_mov", r"v_mov_b(32|64)(_e32|_e64)?$"),
    ("agpr move", r"v_accvgpr_(read|write)"),
    ("select", r"v_cndmask"),
    ("compare", r"v_cmp"),
    ("readlane/readfirstlane", r"v_read(first)?lane|v_writelane"),
    ("other valu", r"v_"),
    ("lds read", r"ds_read"),
    ("lds write", r"ds_write"),
    ("global/buffer", r"(buffer|global|flat|scratch)_"),
    ("s_waitcnt", r"s_waitcnt"),
    ("s_nop", r"s_nop"),
    ("s_barrier", r"s_barrier"),
    ("branch", r"s_c?branch"),
    ("other salu", r"s_"),
]


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "other"


MS, CQ = 7, 13


def main():
    s = open(sys.argv[1]).read()
    sub = sys.argv[2] if len(sys.argv) > 2 else "k_car_bidiag_fusedILi7ELi13"
    name = [n for n in re.findall(r"^(_Z\w+):", s, re.M) if sub in n][0]
    body = s[s.index(name + ":"):]
    body = body[:body.index(".end_amdhsa_kernel")].split("\n")
    # basic blocks: a label line (.LBBn_k: or "; %bb.k:") and the instructions up to the next one; the compiler's
    # comments say which top-level loop a block belongs to ("in Loop: Header=BBn_k Depth=1", "Parent Loop BBn_k Depth=1")
    blocks, cur = [], None
    for l in body:
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
            cur = {"label": l, "lines": []}
            blocks.append(cur)
        elif cur is not None:
            if l.lstrip().startswith(";") and ("Loop" in l):
                cur["label"] += " " + l.strip()
            else:
                cur["lines"].append(l)
    heads = [re.match(r"^\.(LBB\d+_\d+):", b["label"]).group(1) for b in blocks
             if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", b["label"])]
    print(name)
    blk = 0
    for lab in heads:
        key = lab[1:]
        mine = [b for b in blocks if b["label"].startswith("." + lab + ":") or ("Header=" + key + " ") in b["label"] + " "
                or ("Parent Loop " + key + " ") in b["label"] + " "]
        loop = [l for b in mine for l in b["lines"]]
        nbar = sum("s_barrier" in l for l in loop)
        if nbar != 2:
            continue
        ops = [m.group(1) for l in loop for m in [re.match(r"^\s+([a-z_0-9]+)", l)] if m]
        cnt = collections.Counter(classify(o) for o in ops)
        inner = sum(1 for b in mine if "Inner Loop Header: Depth=2" in b["label"])
        tot = len(ops)
        nk, nq = MS - blk, CQ - blk
        want = {"fp64 fma/mul/add": 4 * nk * nq + 4 * nq + 6 * nk + (nk + 1) * 4 + 2 * 20, "dpp mov": 8 * (nk + 1),
                "lds read": 2 * nq + nk + 3 + 16, "lds write": 2 * nq + nk + 2}
        print(f"\nblock S={blk} ({nk} x {nq} live slots): {tot} instructions in the loop body (all paths), {inner} inner (waterfall) loops")
        for k, _ in CLASSES:
            if cnt.get(k):
                w = f"   source asks for ~{want[k]}" if k in want else ""
                print(f"   {k:28s} {cnt[k]:5d}{w}")
        blk += 1


if __name__ == "__main__":
    main()
