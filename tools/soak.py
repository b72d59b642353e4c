"""Seeded soak of the GPU paths against the oracle (the checker): random sizes, data classes, levels,
wrappers, window_bits, mem_level, strategies, section lengths and dest capacities, many streams per
batched call; the oracle runs on the host cores beside it.  usage: soak.py SECONDS [SEED]
(SOAK_BIG=1: few long streams per batch; SOAK_VERBOSE=1: a line per batch before it runs)"""
import os, sys, time, random
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsc_amd import corpus  # (loads libzsc_hip.so but makes no HIP call: the GPU is untouched until zsc_hip_init)

KINDS = ("text", "bitmap", "table", "random", "zero", "runs", "token", "object")
_o = None


def oracle_job(job):
    global _o
    if _o is None:
        from oracle.oracle_py import Oracle
        _o = Oracle()
    kind, n, seed, lvl, wb, ml, strat, mbl, cap = job
    data = corpus.make_buffer(kind, n, seed)
    rc, out, _ = _o.compress(data, lvl, window_bits=wb, mem_level=ml, strategy=strat, max_block_len=mbl,
                             dest_cap=cap, work_len=1 << 21)
    back = _o.uncompress(out, n, wb if wb != 8 else 15) if rc == 0 else None
    return rc, out, back


def inflate_job(job):
    global _o
    if _o is None:
        from oracle.oracle_py import Oracle
        _o = Oracle()
    stream, cap, wb = job
    return _o.uncompress(stream, cap, wb)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    # the oracle's worker processes are forked and warmed up BEFORE this process touches the GPU:
    # a HIP-initialised process must not be forked (its runtime threads, locks and KFD state would
    # be inherited half-alive)
    workers = min(16, os.cpu_count() or 1)
    pool = ProcessPoolExecutor(max_workers=workers)
    list(pool.map(inflate_job, [(b"", 1, 15)] * (4 * workers), chunksize=1))
    import zsc_amd as z
    from oracle.oracle_py import Oracle
    o = Oracle()
    assert z.lib.zsc_hip_init(-1) == 0
    rnd = random.Random(seed)
    t0 = time.time()
    total = streams = bad = batches = 0
    while time.time() - t0 < budget:
        lvl = rnd.choice([1, 2, 3, 4, 5, 6, 6, 6, 7, 8, 9])
        wb = rnd.choice([15, 15, 15, 31, -15, 14, 12, 10, 9, -9, 25])
        ml = rnd.choice([8, 8, 8, 9, 7, 4, 1])
        strat = rnd.choice([0, 0, 0, 1, 4, 2, 3]) if ml >= 8 else rnd.choice([0, 0, 1])
        sections = rnd.random() < 0.5
        jobs = []
        big = os.environ.get("SOAK_BIG") is not None  # few long streams instead of many short ones
        for i in range(rnd.choice([8, 16]) if big else rnd.choice([8, 32, 96])):
            n = rnd.choice([0, 1, 2, 3, 100, 259, 4000, 18432, 18433, 32768, 65535, 65536, 65537, 100000,
                            rnd.randrange(1, 300000), rnd.randrange(1, 2 << 20)])
            if big:
                n = rnd.randrange(1 << 20, 8 << 20)
            kind = rnd.choice(KINDS)
            if sections and n > 1:
                mbl = rnd.choice([rnd.randrange(1, 100), rnd.randrange(100, 5000), rnd.randrange(5000, 70000),
                                  rnd.randrange(60000, 400000), 32768, 65536, 100000])
                if big:
                    mbl = rnd.choice([16384, 32768, 65536, 100000, 262144, 1 << 20, rnd.randrange(20000, 2 << 20)])
                mbl = max(1, min(mbl, n - 1))
                if n // mbl > 400:
                    mbl = n // 400 + 1
            else:
                mbl = max(n, 1)
            bound = o.max_output(n, mbl, lvl, wb, ml)[1]
            cap = rnd.choice([bound, bound, bound, bound + 77, max(1, bound // 2), max(1, bound // 9)])
            jobs.append((kind, n, rnd.randrange(1 << 30), lvl, wb, ml, strat, mbl, cap))
        if os.environ.get("SOAK_VERBOSE"):
            print(f"batch {batches}: level {lvl} wb {wb} mem_level {ml} strategy {strat} sections {sections} "
                  f"{len(jobs)} streams {sum(j[1] for j in jobs) / 1e6:.1f} MB at {time.time() - t0:.0f}s: "
                  + " ".join(f"{j[0]}:{j[1]}/{j[7]}" for j in jobs), flush=True)
        want = pool.map(oracle_job, jobs, chunksize=2)
        bufs = [corpus.make_buffer(j[0], j[1], j[2]) for j in jobs]
        if sections:
            rc, outs, stats = z.compress_sections_batch(bufs, [j[7] for j in jobs], lvl, wb, ml, strat,
                                                        dest_caps=[j[8] for j in jobs])
        else:
            rc, outs, stats = z.compress_batch(bufs, lvl, wb, ml, strat, dest_caps=[j[8] for j in jobs])
        assert rc == 0, (rc, lvl, wb, ml, strat, sections)
        want = list(want)
        good = [(b, w) for b, w in zip(bufs, want) if w[0] == 0]
        if good:
            irc, back, used, ist = z.uncompress_batch([w[1] for _, w in good], [len(b) for b, _ in good],
                                                      window_bits=wb if wb != 8 else 15)
            assert irc == 0
        gi = 0
        for j, b, w, got, st in zip(jobs, bufs, want, outs, stats):
            ok = (st, got) == (w[0], w[1])
            if w[0] == 0:
                ok = ok and (ist[gi], back[gi], used[gi]) == w[2]
                gi += 1
            if not ok:
                bad += 1
                print("MISMATCH", "sections" if sections else "batch", j[:2], j[3:], "gpu", st, len(got), "oracle", w[0], len(w[1]), flush=True)
            total += len(b)
        # the streams the oracle wrote, damaged: the decoder's error paths and resynchronisation
        if good and rnd.random() < 0.5:
            hurt, caps2 = [], []
            for b, w in good:
                c = bytearray(w[1])
                if not c:
                    continue
                for _ in range(rnd.randrange(1, 4)):
                    c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
                if rnd.random() < 0.2:
                    del c[rnd.randrange(len(c)):]
                hurt.append(bytes(c))
                caps2.append(rnd.choice([len(b), len(b) + 50, max(1, len(b) // 2)]))
            iwb = wb if wb != 8 else 15
            want2 = pool.map(inflate_job, [(h, c, iwb) for h, c in zip(hurt, caps2)], chunksize=4)
            irc, outs2, used2, st2 = z.uncompress_batch(hurt, caps2, window_bits=iwb)
            assert irc == 0
            for h, c, o2, u2, s2, w in zip(hurt, caps2, outs2, used2, st2, want2):
                if (s2, o2, u2) != w:
                    bad += 1
                    print("MISMATCH inflate of a damaged stream", len(h), c, iwb, "gpu", s2, len(o2), u2, "oracle", w[0], len(w[1]), w[2], flush=True)
            streams += len(hurt)
        streams += len(jobs)
        batches += 1
        if batches % 10 == 0:
            print(f"  {batches} batches, {streams} streams, {total / 1e6:.0f} MB, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
    print(f"soak seed {seed}: {batches} batches, {streams} streams, {total / 1e6:.1f} MB, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
