"""Collects the counter summaries scripts/gpu_profile_round.sh left under gpurun_out/ into
gpurun_out/TAG_counters.json: per fog schedule ("1": one frame in flight, "4": frames
overlapping) and kernel the mean per dispatch of the VALU wave-instructions, FETCH_SIZE and
WRITE_SIZE (KB), GRBM_GUI_ACTIVE, the kernel's average duration in the kernel trace, plus the hash of the kernel sources the profile was taken on
(bench.py prints the figures only for a build with the same hash)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash  # noqa: E402


def table(path):
    rows = {}
    if not os.path.exists(path):
        return rows
    lines = open(path).read().splitlines()
    head = lines[0].split()[1:]
    for ln in lines[1:]:
        parts = ln.split()
        vals = parts[-len(head):]
        name = " ".join(parts[:-len(head)])
        rows[name] = {h: (None if v == "-" else float(v)) for h, v in zip(head, vals)}
    return rows


def trace_averages(path):
    """kernel -> average duration (ns) in a rocpd_summary.py table of a kernel trace."""
    avg = {}
    if os.path.exists(path):
        for ln in open(path).read().splitlines()[1:]:
            parts = ln.split()
            if len(parts) >= 7:
                avg[" ".join(parts[:-6]).replace("wr::", "")] = float(parts[-4]) * 1e3
    return avg


def main():
    tag = sys.argv[1]
    out = {"source_hash": source_hash(), "tag": tag, "schedules": {}}
    for n in ("1", "4"):
        tr = trace_averages(os.path.join(ROOT, "gpurun_out", f"{tag}_{'kernel' if n == '4' else 'one_frame'}_stats.txt"))
        sq = table(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_sq_{n}.txt"))
        fe = table(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_fetch_{n}.txt"))
        wr = table(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_write_{n}.txt"))
        ks = {}
        for k, d in sq.items():
            get = lambda key: next((v for h, v in d.items() if key.endswith(h) or h.endswith(key[-18:])), None)
            ks[k.replace("wr::", "")] = {
                "valu_wave_instructions": get("SQ_INSTS_VALU"),
                "salu_instructions": get("SQ_INSTS_SALU"),
                "waves": get("SQ_WAVES"),
                "thread_cycles_valu": get("SQ_THREAD_CYCLES_VALU"),
                "gui_active_cycles_all_xcds": get("GRBM_GUI_ACTIVE"),
                "fetch_size_kb": (list(fe.get(k, {}).values()) or [None])[0],
                "write_size_kb": (list(wr.get(k, {}).values()) or [None])[0],
                # average duration of the kernel in the kernel trace of the timed bench command of this
                # schedule (kernels of neighbouring frames and streams overlapping, as in the bench)
                "trace_avg_ns": tr.get(k.replace("wr::", "")),
            }
        out["schedules"][n] = ks
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_counters.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, "hash", out["source_hash"])


if __name__ == "__main__":
    main()
