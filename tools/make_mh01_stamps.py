"""Fixture generator: tests/golden/mh01_stamps.txt.gz = the 3 682 image time stamps (integer nanoseconds, one per line) of EuRoC MH_01 as the
reference ships them for its monocular example (/root/reference/Examples/Monocular/EuRoC_TimeStamps/MH01.txt, read by
Examples/Monocular/mono_euroc.cc:193-199).  DATA, not source: tools/streamed_frontend.cpp --timestamps paces a stream with them
(BASELINE config 3).  Run in the build container, where /root/reference exists."""
import gzip
import os
import sys

SRC = "/root/reference/Examples/Monocular/EuRoC_TimeStamps/MH01.txt"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mh01_stamps.txt.gz")

if __name__ == "__main__":
    stamps = [int(x) for x in open(SRC).read().split()]
    assert len(stamps) == 3682 and all(b > a for a, b in zip(stamps, stamps[1:])), len(stamps)
    with gzip.GzipFile(DST, "wb", mtime=0) as f:
        f.write(("\n".join(str(s) for s in stamps) + "\n").encode())
    d = [(b - a) / 1e6 for a, b in zip(stamps, stamps[1:])]
    print(f"{DST}: {len(stamps)} stamps, {os.path.getsize(DST)} bytes, interval min {min(d):.3f} / median {sorted(d)[len(d)//2]:.3f} / max {max(d):.3f} ms, "
          f"{(stamps[-1] - stamps[0]) / 1e9:.2f} s", file=sys.stderr)
