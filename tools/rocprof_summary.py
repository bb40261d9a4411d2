"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) into a small text table for profiles/.
    python tools/rocprof_summary.py gpurun_out/prof_r1/b64_results.db profiles/r01_kernel_stats_b64.md "<command line>"
Only this library's kernels (namespace aloam::) are listed individually; everything else (torch input generation)
is lumped into one row."""
import sqlite3
import sys


def main(db, out, cmd=""):
    c = sqlite3.connect(db)
    cur = c.execute("select * from top_kernels")
    cols = [d[0] for d in cur.description]
    rows = cur.fetchall()
    mine = [r for r in rows if "aloam::" in r[0]]
    other = [r for r in rows if "aloam::" not in r[0]]
    tot_mine = sum(r[2] for r in mine)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\ncolumns of rocpd `top_kernels`: {cols}\n\n")
        f.write("| kernel | calls | total (us) | avg (us) | % of aloam kernel time |\n|---|---:|---:|---:|---:|\n")
        for r in sorted(mine, key=lambda r: -r[2]):
            name = r[0].split("(")[0].replace("void ", "")
            f.write(f"| `{name}` | {r[1]} | {r[2]:.1f} | {r[2] / r[1]:.2f} | {100 * r[2] / tot_mine:.1f} |\n")
        f.write(f"| all aloam kernels | {sum(r[1] for r in mine)} | {tot_mine:.1f} | | 100 |\n")
        f.write(f"| (torch kernels: synthetic input generation, not timed) | {sum(r[1] for r in other)} | {sum(r[2] for r in other):.1f} | | |\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
