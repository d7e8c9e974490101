"""Reads $PPCSR_DIAG_DUMP (option diag=2): per epoch, how long updates waited, why, and the dependency chain of the worst one.
usage: python tools/diag_chains.py dump.bin [max_epochs]"""
import sys
import numpy as np

WHY = {0: "excl", 1: "barrier", 2: "dup", 3: "W-W", 4: "W-after-R", 5: "R-after-W", 6: "sent-read", 7: "sent-move", 8: "region", 9: "growth", 10: "stamp"}
buf = open(sys.argv[1], "rb").read()
maxe = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
off = 0
ne = 0
tot_hist = np.zeros(64, np.int64)
while off < len(buf) and ne < maxe:
    e0, e1, rounds, viol = np.frombuffer(buf, np.uint64, 4, off)
    off += 32
    cnt = int(e1 - e0)
    tr = np.frombuffer(buf, np.uint32, cnt * 4, off).reshape(cnt, 4)
    off += cnt * 16
    ops = np.frombuffer(buf, np.uint32, cnt * 3, off).reshape(cnt, 3)
    off += cnt * 12
    ne += 1
    wait, why, blk, wlen = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    h = np.bincount(np.minimum(wait, 63), minlength=64)
    tot_hist += h
    worst = int(np.argmax(wait))
    print(f"epoch [{e0},{e1}) rounds {rounds} {'ROLLBACK' if viol else 'done'}: waits mean {wait.mean():.2f} max {wait.max()}; waited>=4: {(wait >= 4).sum()} >=16: {(wait >= 16).sum()}")
    # who are the long waiters?
    lw = wait >= max(4, int(wait.max()) // 2)
    if lw.any():
        srcs, c = np.unique(ops[lw, 0], return_counts=True)
        o = np.argsort(-c)[:12]
        print("   long waiters by src:", " ".join(f"{srcs[i]}:{c[i]}" for i in o), "| last why:", {WHY.get(int(k), k): int(v) for k, v in zip(*np.unique(why[lw], return_counts=True))},
              "| wlen:", {int(k): int(v) for k, v in zip(*np.unique(wlen[lw], return_counts=True))})
    # chain of the worst waiter
    chain = []
    i = worst
    seen = set()
    while 0 <= i < cnt and i not in seen and len(chain) < 24:
        seen.add(i)
        chain.append(f"[{i + int(e0)} src{ops[i, 0]} w{wait[i]} {WHY.get(int(why[i]), why[i])} win{wlen[i]}]")
        b = int(blk[i])
        if wait[i] == 0 or b == 0xFFFFFFFF or b < e0 or b >= e1 or b - int(e0) == i:
            break
        i = b - int(e0)
    print("   worst chain:", " <- ".join(chain))
print("wait histogram (rounds failed -> updates):", {i: int(v) for i, v in enumerate(tot_hist) if v})
