#!/usr/bin/env python3
"""Development probe (CPU only; uses the oracle, hence under tests/): what the lanes of a wave of the GPU's Radau state machine
(csrc/hlm_radau_kernel.h) can be kept busy with, simulated on the oracle's own attempt traces.

The oracle records one byte per implicit attempt (hlm_oracle_job.radau_trace: Newton iterations, outcome, landing on a
forcing boundary).  A wave of 64 lanes is then replayed section by section with the kernel's instruction counts (finish 209
+ set-up 370 per attempt, 346 per Newton iteration after the first; `make -C tiger_hlm_gpu_amd/csrc asm`, tools/isa_blocks.py):
  machine   the kernel's schedule: a trip runs the finish + set-up section once `theta` of the live lanes wait for it, then
            the Newton section for every lane that iterates (theta = 5/8 is the build's HLM_RADAU_SETUP_NUM / _DEN)
  hourly    the same with the lanes re-aligned at every forcing sample boundary (a lane that lands on one waits for its
            wave): neighbours then share the hour's step pattern -- and wait for the slowest of 64
  greedy4   four finer sections (finish, set-up, 3 RHS + residual, solve / update / norm shared by both kinds of lanes),
            always the one most lanes wait for
  shared    the machine with the set-up's own first Newton iteration moved into the shared solve / update / norm tail
Prints active lanes per instruction and the wave's instruction total for SURVEY's cold C3 (DESIGN.md 4.3 quotes them).
usage: python tests/radau_wave_sim.py [ns] [days] [continue]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from tiger_hlm_gpu_amd import synth  # noqa: E402

F_COST, U_COST, N_COST, C_COST = 209.0, 370.0, 346.0, 160.0   # finish, set-up (with its first iteration), Newton section; their shared tail
W = 64


def traces(ns, days, cont):
    sp, y0, cells = synth.model204_params(0, ns), synth.model204_y0(ns), synth.cell_of_system(0, ns)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
    forc = synth.expand_per_stream(pr, t2m, cells)
    tf = days * 1440.0
    r = oracle.run(oracle.UID_204, y0, 0.0, tf, sp=sp, forc=forc, forc_dt_h=dt, forc_nT=nT)
    fl = np.flatnonzero(r["stiff"] == 1)
    sub = dict(sp=sp[fl], forc=np.ascontiguousarray(forc[:, fl]), forc_dt_h=dt, forc_nT=nT, radau_only=True, radau_trace_cap=20000)
    if cont:
        rr = oracle.run(oracle.UID_204, y0[fl], 0.0, tf, continue_from=(r["t_reached"][fl], r["y_reached"][fl]), **sub)
        start = r["t_reached"][fl]
    else:
        rr = oracle.run(oracle.UID_204, y0[fl], 0.0, tf, **sub)
        start = np.zeros(len(fl))
    return rr["radau_trace"], rr["n_attempt"].astype(int), start


def machine(buf, ln, start, theta=0.625, hourly=False, shared=False):
    """the kernel's two-section machine; returns (lane-instructions, wave-instructions)"""
    order, nxt = list(range(len(ln))), 0
    sysi = np.full(W, -1); pos = np.zeros(W, int); rem = np.zeros(W, int)
    phase = np.full(W, 2); ended = np.zeros(W, int); hb = np.zeros(W, int)   # phase 0 Newton, 1 set-up, 2 done

    def refill(lane):
        nonlocal nxt
        if nxt < len(order):
            sysi[lane] = order[nxt]; nxt += 1; pos[lane] = 0; phase[lane] = 1; ended[lane] = 0
            hb[lane] = int(start[sysi[lane]] // 60)
            return True
        phase[lane] = 2
        return False
    for lane in range(W):
        refill(lane)
    lane_work = wave_cost = 0.0
    while True:
        live = phase != 2
        if not live.any():
            if hourly and nxt < len(order):   # (aligned lanes take the next 64 systems together)
                for lane in range(W):
                    refill(lane)
                continue
            break
        hold = np.zeros(W, bool)
        if hourly:
            hold = live & (phase == 1) & (ended == 0) & (hb > hb[live].min())
        ms, mn = live & (phase == 1) & ~hold, live & (phase == 0)
        n_s, n_n = ms.sum(), mn.sum()
        newton_now = mn.copy()
        tail = False
        if n_s > 0 and (n_n == 0 or n_s >= (n_s + n_n) * theta):
            fin = ms & (ended == 1)
            if fin.any():
                wave_cost += F_COST; lane_work += F_COST * fin.sum()
            setup = []
            for lane in np.flatnonzero(ms):
                if ended[lane]:
                    ended[lane] = 0
                    if buf[sysi[lane], pos[lane] - 1] & 16: hb[lane] += 1
                if pos[lane] >= ln[sysi[lane]]:
                    if hourly: phase[lane] = 2
                    else: refill(lane)
                    continue
                if hourly and hb[lane] > hb[phase != 2].min():
                    continue
                setup.append(lane)
            if setup:
                wave_cost += U_COST - (C_COST if shared else 0.0); lane_work += U_COST * len(setup); tail = True
                for lane in setup:
                    its = buf[sysi[lane], pos[lane]] & 15; pos[lane] += 1
                    if its <= 1: ended[lane] = 1
                    else: rem[lane] = its - 1; phase[lane] = 0
        mn = (phase == 0) & (newton_now if shared else True)   # (shared tail: a lane's first RHS iteration is the next trip's)
        if mn.any():
            wave_cost += N_COST - (C_COST if shared else 0.0); lane_work += N_COST * mn.sum(); tail = True
            rem[mn] -= 1
            done = mn & (rem == 0)
            phase[done] = 1; ended[done] = 1
        if shared and tail: wave_cost += C_COST
    return lane_work, wave_cost


def greedy4(buf, ln, overhead=12.0):
    cost = {"F": F_COST, "U": U_COST - C_COST, "N": N_COST - C_COST, "C": C_COST}
    nxt = 0
    sysi = np.full(W, -1); pos = np.zeros(W, int); rem = np.zeros(W, int); st = ["X"] * W
    for lane in range(W):
        if nxt < len(ln): sysi[lane] = nxt; nxt += 1; st[lane] = "U"
    lane_work = wave_cost = 0.0
    while True:
        cnt = {k: sum(1 for x in st if x == k) for k in "FUNC"}
        if sum(cnt.values()) == 0: break
        sec = max("FUNC", key=lambda k: cnt[k])
        wave_cost += cost[sec] + overhead; lane_work += cost[sec] * cnt[sec]
        for lane in range(W):
            if st[lane] != sec: continue
            if sec == "F":
                if pos[lane] >= ln[sysi[lane]]:
                    if nxt < len(ln): sysi[lane] = nxt; nxt += 1; pos[lane] = 0; st[lane] = "U"
                    else: st[lane] = "X"
                else: st[lane] = "U"
            elif sec == "U":
                rem[lane] = buf[sysi[lane], pos[lane]] & 15; pos[lane] += 1; st[lane] = "C"
            elif sec == "N": st[lane] = "C"
            else:
                rem[lane] -= 1; st[lane] = "N" if rem[lane] > 0 else "F"
    return lane_work, wave_cost


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    days = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    cont = len(sys.argv) > 3 and sys.argv[3] == "continue"
    buf, ln, start = traces(ns, days, cont)
    its = np.concatenate([buf[s, :ln[s]] & 15 for s in range(len(ln))])
    kinds = np.concatenate([buf[s, :ln[s]] >> 4 for s in range(len(ln))])
    print(f"{len(ln)} flagged of {ns}, {'continue' if cont else 'restart'}: {ln.mean():.0f} implicit attempts per system, "
          f"{(kinds & 2 > 0).mean():.3f} failed Newton, {(kinds & 4 > 0).mean():.3f} rejected; iterations per attempt "
          f"{its.mean():.2f} ({', '.join(f'{k}: {np.mean(its == k):.2f}' for k in range(2, 6))})")
    rows = [("machine, theta 1/2", dict(theta=0.5)), ("machine, theta 5/8 (the build)", dict(theta=0.625)),
            ("machine, theta 3/4", dict(theta=0.75)), ("hourly re-alignment", dict(hourly=True)),
            ("shared solve / update / norm tail", dict(shared=True))]
    for name, kw in rows:
        lw, wc = machine(buf, ln, start, **kw)
        print(f"  {name:36s} {lw / wc:5.1f} lanes per instruction, {wc:.4g} instructions")
    lw, wc = greedy4(buf, ln)
    print(f"  {'four sections, most lanes first':36s} {lw / wc:5.1f} lanes per instruction, {wc:.4g} instructions")


if __name__ == "__main__":
    main()
