#!/usr/bin/env python3
"""Offline LDS bank-conflict calculator for gfx950 (MI355X).

Banking rules follow /opt/skills/guides/MI355X_MICROARCH.md §LDS:
  * a wave64 DS access is serviced in fixed lane groups, one LDS cycle per group when
    conflict free; each extra distinct dword address on a busy bank adds a cycle;
  * bank = (addr/4) % 64 for ds_read_b64 / ds_read_b128 / ds_read_b64_tr_b16,
    (addr/4) % 32 for ds_read_b32 and every ds_write.

Used while designing the LDS images in clip_dplm_amd/csrc (no GPU needed).
"""
from __future__ import annotations

GROUPS = {
    "read_b128": [
        [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
        [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
        [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
    ],
    "read_b64": [list(range(0, 32)), list(range(32, 64))],
    "read_tr_b64": [list(range(0, 32)), list(range(32, 64))],
    "read_b32": [list(range(0, 32)), list(range(32, 64))],
    "write_b32": [list(range(0, 32)), list(range(32, 64))],
    "write_b64": [list(range(16 * g, 16 * g + 16)) for g in range(4)],
    "write_b128": [list(range(8 * g, 8 * g + 8)) for g in range(8)],
}
WIDTH = {"read_b128": 16, "read_b64": 8, "read_tr_b64": 8, "read_b32": 4,
         "write_b32": 4, "write_b64": 8, "write_b128": 16}
NBANKS = {"read_b128": 64, "read_b64": 64, "read_tr_b64": 64, "read_b32": 32,
          "write_b32": 32, "write_b64": 32, "write_b128": 32}


def cycles(kind: str, addrs: list[int]) -> tuple[int, int]:
    """Return (cycles, ideal_cycles) for one wave instruction with per-lane byte addresses."""
    assert len(addrs) == 64
    total = 0
    for grp in GROUPS[kind]:
        per_bank: dict[int, set[int]] = {}
        for lane in grp:
            a = addrs[lane]
            for d in range(WIDTH[kind] // 4):
                dw = a // 4 + d
                per_bank.setdefault(dw % NBANKS[kind], set()).add(dw)
        total += max(len(v) for v in per_bank.values())
    return total, len(GROUPS[kind])


def report(name: str, kind: str, addrs: list[int]) -> None:
    c, ideal = cycles(kind, addrs)
    print(f"{name:48s} {kind:12s} cycles={c:3d} ideal={ideal} -> {c / ideal:.1f}x")


if __name__ == "__main__":
    # ---- GEMM NT operand tile: rows of BK bf16, ds_read_b128 fragments (row = lane&15, chunk = lane>>4)
    for bk, sw in ((64, lambda r: (r >> 1) & 7), (32, lambda r: (r >> 2) & 3)):
        rowb = bk * 2
        for kk in range(bk // 32):
            addrs = []
            for l in range(64):
                r, c = l & 15, (l >> 4) + 4 * kk
                addrs.append(r * rowb + ((c ^ sw(r)) * 16))
            report(f"gemm_nt BK={bk} kk={kk} swizzled", "read_b128", addrs)
            addrs = [(l & 15) * rowb + ((l >> 4) + 4 * kk) * 16 for l in range(64)]
            report(f"gemm_nt BK={bk} kk={kk} linear", "read_b128", addrs)
        # staging write: thread t -> row t>>log2(bk/8), chunk t&(bk/8-1)
        cpr = bk // 8
        addrs = []
        for l in range(64):
            r, c = l // cpr, l % cpr
            addrs.append(r * rowb + ((c ^ sw(r)) * 16))
        report(f"gemm_nt BK={bk} stage write", "write_b128", addrs)

    # ---- transposed reads (wgrad / attention V): tile [rows][cols] bf16, row stride S bytes
    for stride in (256, 288, 96, 160, 64 + 32, 128 + 32, 192 + 32, 256 + 32, 320 + 32):
        addrs = []
        for l in range(64):
            g, i = l >> 4, l & 15
            q, p = i >> 2, i & 3
            addrs.append((4 * g + q) * stride + 8 * p)
        report(f"tr_read block0 stride={stride}", "read_tr_b64", addrs)

    # ---- attention K tile row reads: [key][DP] bf16 with padded stride
    for dp in (32, 64, 96, 128, 160):
        for pad in (0, 16, 32):
            stride = dp * 2 + pad
            addrs = [(l & 15) * stride + (l >> 4) * 16 for l in range(64)]
            report(f"attn row-read DP={dp} stride={stride}", "read_b128", addrs)
