#!/usr/bin/env python3
"""GPU timeline summary of a rocprofv3 --kernel-trace database (rocpd sqlite): per stream / queue busy time, union busy
time, idle time and the largest gaps inside the steady-state window (the last `steps` train steps).
    usage: timeline.py <results.db> <out.json> [steps_in_window]"""
import json
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e, gaps = 0, None, None, []
    for s, e in iv:
        if cur_s is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            gaps.append((s - cur_e, cur_e, s))
            tot += cur_e - cur_s
            cur_s, cur_e = s, e
    if cur_s is not None:
        tot += cur_e - cur_s
    return tot, gaps


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("PRAGMA table_info(kernels)")]
    key = [c for c in ("stream_id", "queue_id") if c in cols]
    sel = ", ".join(["name", "start", "end"] + key)
    rows = list(cur.execute(f"select {sel} from kernels order by start"))
    # the train steps are delimited by the fused Adam launch
    adam = [r[1] for r in rows if "adam_kernel" in r[0]]
    res = {"columns": cols, "n_kernels": len(rows), "adam_launches": len(adam)}
    if len(adam) > steps:
        t0, t1 = adam[-steps - 1], adam[-1]
        win = [r for r in rows if t0 <= r[1] < t1]
        res["window_ms"] = (t1 - t0) / 1e6
        res["ms_per_step"] = (t1 - t0) / 1e6 / steps
        tot, gaps = union([(r[1], r[2]) for r in win])
        res["union_busy_ms_per_step"] = tot / 1e6 / steps
        res["idle_ms_per_step"] = ((t1 - t0) - tot) / 1e6 / steps
        res["kernel_time_sum_ms_per_step"] = sum(r[2] - r[1] for r in win) / 1e6 / steps
        res["launches_per_step"] = len(win) / steps
        gaps.sort(reverse=True)
        res["gaps_over_5us_per_step"] = sum(1 for g in gaps if g[0] > 5000) / steps
        res["gap_time_over_5us_ms_per_step"] = sum(g[0] for g in gaps if g[0] > 5000) / 1e6 / steps
        res["gap_time_under_5us_ms_per_step"] = sum(g[0] for g in gaps if g[0] <= 5000) / 1e6 / steps
        names = {}
        for r in win:
            pass
        # which kernel FOLLOWS the largest gaps
        big = []
        starts = {r[1]: r[0] for r in win}
        for g in gaps[:25]:
            big.append({"gap_us": g[0] / 1e3, "before": starts.get(g[2], "?")[:70]})
        res["largest_gaps"] = big
        # when only the side stream runs (the compute stream waits): which side kernels, and what the compute stream resumes with
        if key:
            idx0 = 3
            streams = sorted(set(r[idx0] for r in win))
            if len(streams) >= 2:
                main_id = max(streams, key=lambda q: sum(r[2] - r[1] for r in win if r[idx0] == q))
                main_iv = sorted((r[1], r[2]) for r in win if r[idx0] == main_id)
                _, main_gaps = union(main_iv)
                side = [r for r in win if r[idx0] != main_id]
                starts_main = {r[1]: r[0] for r in win if r[idx0] == main_id}
                waits = []
                for glen, gs, ge in main_gaps:
                    if glen < 20000:
                        continue
                    active = {}
                    for r in side:
                        ov = min(ge, r[2]) - max(gs, r[1])
                        if ov > 0:
                            active[r[0][:60]] = active.get(r[0][:60], 0) + ov
                    waits.append({"main_idle_us": glen / 1e3, "resumes_with": starts_main.get(ge, "?")[:60],
                                  "side_busy_us": {k: round(v / 1e3, 1) for k, v in sorted(active.items(), key=lambda kv: -kv[1])[:3]}})
                waits.sort(key=lambda w: -w["main_idle_us"])
                res["main_stream_idle_over_20us_ms_per_step"] = sum(w["main_idle_us"] for w in waits) / 1e3 / steps
                res["main_stream_waits"] = waits[:14]
        for k in key:
            idx = 3 + key.index(k)
            per = {}
            for r in win:
                per.setdefault(str(r[idx]), []).append((r[1], r[2]))
            res[f"busy_ms_per_step_by_{k}"] = {q: union(iv)[0] / 1e6 / steps for q, iv in per.items()}
    # per-launch trace of the LAST step of the window (start relative to the step's first launch)
    if len(adam) > steps:
        last = [r for r in rows if adam[-2] <= r[1] < adam[-1]]
        if last:
            base = last[0][1]
            with open(out.replace(".json", "") + "_trace.csv", "w") as f:
                f.write("start_us,dur_us,stream,kernel\n")
                for r in last:
                    nm = r[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60].replace(",", ";")
                    f.write(f"{(r[1] - base) / 1e3:.1f},{(r[2] - r[1]) / 1e3:.1f},{r[3] if len(r) > 3 else 0},{nm}\n")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("largest_gaps", "columns", "main_stream_waits")}, indent=1))
    for w in res.get("main_stream_waits", [])[:10]:
        print(w)
    for g in res.get("largest_gaps", [])[:12]:
        print(g)


if __name__ == "__main__":
    main()
