"""Which threads burn the CPU in a multi-rank run?  Starts `python bench.py <args>` in the background and, while it
runs, samples utime + stime of every thread of every rank process (/proc/<pid>/task/*/stat) twice, some seconds
apart: prints the busiest threads with their names and kernel wait channels.
  gpurun -- 'python tools/attic/thread_cpu_probe.py 30 6 --gpus 2 --ranks-share-gpu --rows 131072 --steps 400 ...'
  (first argument: seconds to wait before the first sample, second: seconds between the samples)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def threads_of(pid):
    out = {}
    try:
        for tid in os.listdir("/proc/%d/task" % pid):
            try:
                st = open("/proc/%d/task/%s/stat" % (pid, tid)).read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                wchan = open("/proc/%d/task/%s/wchan" % (pid, tid)).read().strip()
                out[int(tid)] = (name, int(f[11]) + int(f[12]), f[0], wchan)
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return out


def descendants(pid):
    kids, todo = [], [pid]
    while todo:
        p = todo.pop()
        try:
            for t in os.listdir("/proc/%d/task" % p):
                ch = open("/proc/%d/task/%s/children" % (p, t)).read().split()
                for c in ch:
                    kids.append(int(c))
                    todo.append(int(c))
        except OSError:
            pass
    return kids


def main():
    wait_s, gap_s = float(sys.argv[1]), float(sys.argv[2])
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[3:], cwd=ROOT,
                         stdout=open("/tmp/thread_probe.out", "w"), stderr=open("/tmp/thread_probe.err", "w"))
    time.sleep(wait_s)
    hz = os.sysconf("SC_CLK_TCK")
    for rnd in range(2):
        pids = [p.pid] + descendants(p.pid)
        a = {pid: threads_of(pid) for pid in pids}
        time.sleep(gap_s)
        b = {pid: threads_of(pid) for pid in pids}
        rows = []
        for pid in pids:
            for tid, (name, t1, state, wchan) in b[pid].items():
                t0 = a[pid].get(tid, (name, t1, "", ""))[1]
                rows.append(((t1 - t0) / hz / gap_s, pid, tid, name, state, wchan))
        rows.sort(reverse=True)
        print("sample %d: %d processes, %d threads, %.1f CPUs busy in total" % (rnd, len(pids), len(rows), sum(r[0] for r in rows)))
        for r in rows[:24]:
            print("   %5.2f cpu  pid %d tid %d  %-18s state %s wchan %s" % r)
        per = {}
        for r in rows:
            key = (r[1], r[5] if r[0] < 0.05 else "busy")
            per.setdefault(key, [0, 0.0])
            per[key][0] += 1
            per[key][1] += r[0]
        for (pid, w), (cnt, cpu) in sorted(per.items()):
            print("   pid %d  %-28s %4d threads %6.2f cpu" % (pid, w, cnt, cpu))
        sys.stdout.flush()
    print("bench still running:", p.poll() is None)
    print(open("/tmp/thread_probe.err").read()[-1500:])
    p.terminate()
    try:
        p.wait(timeout=20)
    except subprocess.TimeoutExpired:
        p.kill()


if __name__ == "__main__":
    main()
