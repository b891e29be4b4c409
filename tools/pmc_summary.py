#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 --pmc passes written by tools/bench_pmc.sh.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; write bytes = WRITE_SIZE * 1024.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs): rocprofv3 sums GRBM_GUI_ACTIVE over
the 8 XCDs (checked: GUI_ACTIVE / 8 / kernel duration = 2.16 GHz), the MFMA busy cycles over all 1024 SIMDs.
Durations under --pmc are serialised / inflated and are NOT used; pair the bytes with the kernel-trace durations."""
import collections
import csv
import glob
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(void )?([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    s = m.group(2) if m else name
    return s[:70]


def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            out[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                cnt[k] += 1
    return out, cnt


def is_128(k):
    """the 128x128 tile class: gemm_x3_kernel (bf16 x 3 split, default) and gemm_kernel<128, 128, ...> (native fp32)"""
    return k.startswith('gemm_kernel<128, 128') or k.startswith('gemm_x3_kernel')


def by_launch(root, dst):
    """128x128-class launches grouped by (kernel variant, grid size): grid = tiles x split-K x workgroup size identifies the
    shape; read / write bytes per dispatch at the L2<->fabric boundary (join with TWOG_BENCH_GEMM_DETAIL=1 of bench.py)."""
    # (the two passes are separate runs of the bench and may execute different numbers of settling steps: each counter is
    # divided by the dispatches of ITS pass)
    agg = collections.defaultdict(lambda: [set(), 0.0, 0.0, set()])
    for sub, col, mul in (('f', 1, 2.0 * 1024), ('w', 2, 1024.0)):
        for f in glob.glob(root + '/' + sub + '/*/*_counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if not is_128(k) or r['Counter_Name'] not in ('FETCH_SIZE', 'WRITE_SIZE'):
                    continue
                a = agg[(k, int(r['Grid_Size']), int(r['Workgroup_Size']))]
                a[0 if sub == 'f' else 3].add(r['Dispatch_Id'])
                a[col] += float(r['Counter_Value']) * mul
    with open(dst, 'w', newline='') as fo:
        wri = csv.writer(fo)
        wri.writerow(['kernel', 'workgroups', 'dispatches', 'read_MB_per_dispatch', 'write_MB_per_dispatch'])
        for (k, grid, wg), (ids, rd, wr, idw) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            n, nw = max(len(ids), 1), max(len(idw), 1)
            wri.writerow([k, grid // max(wg, 1), n, round(rd / n / 1e6, 1), round(wr / nw / 1e6, 1)])
    return agg


HOST_MIN_WORKGROUPS = 400   # launches of the class issued by the host composition (the ones bench.py's events time) have
                            # >= 480 workgroups; the segment chain's per-step projection launch inside the library has 240


def host_issued(agg):
    """(read bytes per launch + written bytes per launch, dispatches) over the launches with >= HOST_MIN_WORKGROUPS
    workgroups, each counter divided by the dispatches of its own pass."""
    rd = wr = 0.0
    n = nw = 0
    for (k, grid, wg), (ids, r_, w_, idw) in agg.items():
        if grid // max(wg, 1) >= HOST_MIN_WORKGROUPS:
            rd += r_
            wr += w_
            n += len(ids)
            nw += len(idw)
    return rd / max(n, 1) + wr / max(nw, 1), n, nw, rd + wr


def main():
    root, dst = sys.argv[1], sys.argv[2]
    agg128 = by_launch(root, sys.argv[4]) if len(sys.argv) > 4 else None
    f, nf = load(root + '/f')
    w, nw = load(root + '/w')
    s, _ = load(root + '/s')
    rows = []
    for k in nf:
        rd = 2.0 * f[k].get('FETCH_SIZE', 0.0) * 1024
        wr = w.get(k, {}).get('WRITE_SIZE', 0.0) * 1024
        sq = s.get(k, {})
        gui = sq.get('GRBM_GUI_ACTIVE', 0.0)
        mfma = sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        rows.append(dict(kernel=k, dispatches=nf[k], hbm_read_bytes=int(rd), hbm_write_bytes=int(wr),
                         hbm_bytes_per_dispatch=int(rd / max(nf[k], 1) + wr / max(nw.get(k, 0), 1)),
                         mfma_util_pct=round(100.0 * mfma / (gui / 8 * 256 * 4), 2) if gui else 0.0,
                         sq_wait_any_frac=round(sq.get('SQ_WAIT_ANY', 0.0) / sq['SQ_WAVE_CYCLES'], 3) if sq.get('SQ_WAVE_CYCLES') else 0.0,
                         sq_wait_inst_frac=round(sq.get('SQ_WAIT_INST_ANY', 0.0) / sq['SQ_WAVE_CYCLES'], 3) if sq.get('SQ_WAVE_CYCLES') else 0.0))
    rows.sort(key=lambda r: -(r['hbm_read_bytes'] + r['hbm_write_bytes']))
    with open(dst, 'w', newline='') as fo:
        wri = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        wri.writeheader()
        wri.writerows(rows)
    big = [r for r in rows if is_128(r['kernel'])]
    if big and len(sys.argv) > 3:
        import json
        nd = sum(r['dispatches'] for r in big)
        ndw = max(sum(nw.get(r['kernel'], 0) for r in big), 1)
        by = sum(r['hbm_read_bytes'] + r['hbm_write_bytes'] for r in big)
        per_launch = sum(r['hbm_read_bytes'] for r in big) / nd + sum(r['hbm_write_bytes'] for r in big) / ndw
        note = 'all launches of the class'
        if agg128 is not None:   # the launches bench.py's roofline covers: issued by the host composition
            per_launch, nd, ndw, by = host_issued(agg128)
            note = (f'launches with >= {HOST_MIN_WORKGROUPS} workgroups = the host-issued ones that bench.py times with events; the '
                    'segment chain\'s per-step 240-tile launch of the class is listed in the by-launch csv')
        json.dump(dict(kernel='128x128 tile class: gemm_x3_kernel<*> + gemm_kernel<128,128,*>', dispatches=nd, dispatches_write_pass=ndw,
                       hbm_bytes_total=by, hbm_bytes_per_launch=per_launch, launches=note,
                       source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`'
                              ' (tools/bench_pmc.sh); read bytes = 2 * FETCH_SIZE KiB (gfx950 correction)'),
                  open(sys.argv[3], 'w'), indent=1)
    tot = sum(r['hbm_read_bytes'] + r['hbm_write_bytes'] for r in rows)
    print('total HBM bytes (2 bench steps incl. warmup + setup):', tot)
    for r in rows[:14]:
        print(r)


if __name__ == '__main__':
    main()
