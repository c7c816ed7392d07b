"""How fast can the box read a slot file that is NOT in the page cache?  (Decides whether O_DIRECT reads into the pinned ring are
worth having: DESIGN.md section 9, "Next".)  Writes a file, evicts it (fsync + POSIX_FADV_DONTNEED: no root needed), then
reads it back with 1..16 threads, buffered and O_DIRECT, 8 MiB requests into page-aligned buffers.
Usage: disk_probe.py [dir] [GiB]"""
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

d = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
size = int(gib * (1 << 30)) // (8 << 20) * (8 << 20)
path = os.path.join(d, "cp2_disk_probe.bin")
REQ = 8 << 20


def evict():
    fd = os.open(path, os.O_RDONLY)
    os.fsync(fd)
    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    os.close(fd)


def read_all(threads, direct):
    flags = os.O_RDONLY | (os.O_DIRECT if direct else 0)
    try:
        fds = [os.open(path, flags) for _ in range(threads)]
    except OSError as e:
        return None, "open failed: %s" % e
    bufs = [mmap.mmap(-1, REQ) for _ in range(threads)]          # anonymous mappings are page aligned
    n_req = size // REQ

    def work(t):
        got = 0
        for r in range(t, n_req, threads):
            got += os.preadv(fds[t], [bufs[t]], r * REQ)
        return got
    t0 = time.perf_counter()
    try:
        with ThreadPoolExecutor(threads) as ex:
            total = sum(ex.map(work, range(threads)))
    except OSError as e:
        return None, "read failed: %s" % e
    dt = time.perf_counter() - t0
    for fd in fds:
        os.close(fd)
    return total / dt / 1e9, "ok" if total == size else "short read %d" % total


print("directory %s, file of %.1f GiB, %d MiB requests" % (d, size / 2**30, REQ >> 20))
try:
    st = os.statvfs(d)
    print("filesystem: free %.0f GiB" % (st.f_bavail * st.f_frsize / 2**30), "| mount:", [l.split()[:3] for l in open("/proc/mounts") if l.split()[1] in (d, "/")][:2])
except Exception as e:
    print("statvfs:", e)
t0 = time.perf_counter()
with open(path, "wb") as f:
    block = os.urandom(REQ)
    for _ in range(size // REQ):
        f.write(block)
    f.flush()
    os.fsync(f.fileno())
print("write + fsync: %.2f GB/s" % (size / (time.perf_counter() - t0) / 1e9), flush=True)
try:
    for direct in (False, True):
        for threads in (1, 4, 8, 16):
            evict()
            gbps, msg = read_all(threads, direct)
            warm = None
            if not direct and gbps:
                warm, _ = read_all(threads, False)           # second pass: page cache
            print("%-9s %2d threads: cold %s GB/s%s  (%s)" % ("O_DIRECT" if direct else "buffered", threads, "%.2f" % gbps if gbps else "-",
                                                             ", warm %.2f GB/s" % warm if warm else "", msg), flush=True)
finally:
    os.remove(path)
