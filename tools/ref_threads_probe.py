#!/usr/bin/env python3
"""How fast can the host run the REFERENCE's rounds >= 12?  bench.py's cpu_baseline uses `-p 64` unpinned; this probe times
the same binary (oracle/_ref/rsem-run-em) on the same generated sample (5 % of configs[2]) at other thread counts and with
the threads pinned to physical cores (one hardware thread per core; one socket / both sockets), each for a bounded time:
ms per round from the arrival times of its 'ROUND =' lines, like bench.py.  Prints one JSON object.

usage: ref_threads_probe.py [seconds_of_rounds_per_setting = 14]"""
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = dict(frac=0.05, M=200_000, iso="5-16")  # = bench.py's CPU_SAMPLE["C3"]


def topology():
    """-> {package: [one cpu per physical core, ...]} from sysfs"""
    cores = {}
    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
        cpu = int(os.path.basename(d)[3:])
        try:
            pkg = int(open(d + "/topology/physical_package_id").read())
            sib = open(d + "/topology/thread_siblings_list").read().strip()
        except OSError:
            continue
        first = int(sib.replace("-", ",").split(",")[0])
        if first == cpu:
            cores.setdefault(pkg, []).append(cpu)
    return {k: sorted(v) for k, v in sorted(cores.items())}


def main():
    steady_s = float(sys.argv[1]) if len(sys.argv) > 1 else 14.0
    cs = SAMPLE
    gen = os.path.join(ROOT, "tools", "bin", "gen_temp")
    ref_em = os.path.join(ROOT, "oracle", "_ref", "rsem-run-em")
    ref_idx = os.path.join(ROOT, "oracle", "_ref", "rsem-build-read-index")
    d = tempfile.mkdtemp(prefix="rsem_probe_", dir="/tmp")
    n_reads = int(os.environ.get("PROBE_READS", int(50_000_000 * cs["frac"] / 0.95)))  # (the variable: for trying the script out)
    out = subprocess.run([gen, d, str(n_reads), str(cs["M"]), "3", "20250925", "100", "nosam", cs["iso"]], stdout=subprocess.PIPE, text=True, check=True).stdout
    subprocess.run([ref_idx, "32", "1", "1"] + [os.path.join(d, "temp", r) for r in ("s_alignable_1.fq", "s_alignable_2.fq")], stdout=subprocess.DEVNULL, check=True)
    nhits = int(out.split("nHits=")[1].split()[0])
    topo = topology()
    pk = sorted(topo)
    one_socket = topo[pk[0]] if pk else []
    all_cores = [c for p in pk for c in topo[p]]
    settings = [("-p 64, unpinned (bench.py's cpu_baseline)", 64, None)]
    if one_socket:
        settings.append(("-p %d, pinned to the %d physical cores of socket %d" % (len(one_socket), len(one_socket), pk[0]), len(one_socket), one_socket))
    if len(pk) > 1:
        settings.append(("-p %d, pinned to one hardware thread per physical core, both sockets" % len(all_cores), len(all_cores), all_cores))
        settings.append(("-p %d, unpinned" % len(all_cores), len(all_cores), None))
    res = {"sample": "5 %% of configs[2]: %d alignments, %d transcripts" % (nhits, cs["M"]), "host": {"cpus": os.cpu_count(), "packages": len(pk), "physical_cores": len(all_cores)},
           "seconds_of_rounds_per_setting": steady_s, "settings": []}
    for name, p, cpus in settings:
        cmd = [ref_em, os.path.join(d, "ref"), "3", os.path.join(d, "s"), os.path.join(d, "temp", "s"), os.path.join(d, "stat", "s"), "-p", str(p)]
        if cpus:
            cmd = ["taskset", "-c", ",".join(map(str, cpus))] + cmd
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        marks = []
        for line in proc.stdout:
            now = time.perf_counter()
            if line.startswith("ROUND ="):
                marks.append((int(line.split(",")[0].split("=")[1]), now))
            late = [m for m in marks if m[0] >= 12]
            if (late and now - late[0][1] > steady_s) or now - t0 > 120:
                break
        proc.kill()
        proc.wait()
        late = [m for m in marks if m[0] >= 12]
        row = {"setting": name, "threads": p}
        if len(late) >= 3:
            per = (late[-1][1] - late[0][1]) / (late[-1][0] - late[0][0])
            row.update({"ms_per_round": per * 1e3, "rounds_timed": late[-1][0] - late[0][0], "read_alignments_per_s": nhits / per,
                        "startup_s": marks[0][1] - t0, "rounds_1_11_s": late[0][1] - marks[0][1]})
        else:
            row["error"] = "fewer than 3 rounds >= 12 seen"
        res["settings"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    subprocess.run(["rm", "-rf", d])
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
