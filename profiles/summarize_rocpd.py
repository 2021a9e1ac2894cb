"""Turn a rocprofv3 rocpd SQLite result (kernel-trace) into the per-kernel stats table that
`rocprofv3 --stats` prints: calls, total/avg/min/max duration, share of GPU kernel time."""
import sqlite3
import sys


def main(db_path, out_path=None, top=40):
    db = sqlite3.connect(db_path)
    rows = db.execute("""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    lines = [f'# source: {db_path}', f'# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches',
             '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for name, calls, tot, avg, mn, mx in rows[:top]:
        short = name if len(name) < 110 else name[:107] + '...'
        lines.append(f'| `{short}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |')
    text = '\n'.join(lines) + '\n'
    if out_path:
        open(out_path, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:3])
