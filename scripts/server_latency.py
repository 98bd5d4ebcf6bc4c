#!/usr/bin/env python3
"""Query latency of the resident bgt-server (images in HBM) next to one `bgt view` process per query -- this repo's CLI and
the compiled reference -- on a synthetic database.  Run on the GPU box: python scripts/server_latency.py [samples] [sites]"""
import os
import socket
import subprocess
import sys
import tempfile
import time
import urllib.parse
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bgt_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "bgt")
n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bgt_amd", "host")])
tmp = tempfile.mkdtemp()
db = os.path.join(tmp, "db")
subprocess.check_call([os.path.join(BIN, "bgt"), "synth", db, str(n_samples), str(n_sites), "2"], stdout=subprocess.DEVNULL)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
t0 = time.perf_counter()
srv = subprocess.Popen([os.path.join(BIN, "bgt-server"), "-p", str(port), "-m", "4000000000", db], stderr=open(os.path.join(tmp, "server.log"), "w"),
                       env=dict(os.environ, BGS_TRACE="1", BGT_TRACE="1", BGTH_TRACE="1"))
while True:
    try:
        socket.create_connection(("127.0.0.1", port), timeout=1).close()
        break
    except OSError:
        assert srv.poll() is None
        time.sleep(0.02)
print("database %d samples x %d sites; server ready (images resident) after %.2f s" % (n_samples, n_sites, time.perf_counter() - t0))
mid = 1000 + 10 * (n_sites // 2)
QUERIES = [   # (label, server query, view arguments)
    ("region of 100 sites, AC/AN", "C&r=11:%d-%d" % (mid, mid + 999), ["-G", "-C", "-r", "11:%d-%d" % (mid, mid + 999)]),
    ("region of 10,000 sites, filter", "f=AC%%3E0&r=11:%d-%d" % (mid, mid + 99999), ["-G", "-f", "AC>0", "-r", "11:%d-%d" % (mid, mid + 99999)]),
    ("two groups over 10,000 sites", "s=" + urllib.parse.quote('pop=="A"') + "&s=" + urllib.parse.quote('pop=="B"') + "&f=(AC1%%3E0.and.AC2==0)&r=11:%d-%d" % (mid, mid + 99999),
     ["-G", "-s", 'pop=="A"', "-s", 'pop=="B"', "-f", "AC1>0&&AC2==0", "-r", "11:%d-%d" % (mid, mid + 99999)]),
    ("genotypes of 20 samples, 1,000 sites", "g&s=idx%%3C20&r=11:%d-%d" % (mid, mid + 9999), ["-s", "idx<20", "-r", "11:%d-%d" % (mid, mid + 9999)]),
    ("records 500,001..500,100 by number", "C&i=%d&n=99" % (n_sites // 2 + 1), ["-G", "-C", "-i", str(n_sites // 2 + 1), "-n", "100"]),
]


def best(f, n=5):
    ts, out = [], None
    for _ in range(n):
        t = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t)
    return min(ts), out


for label, q, va in QUERIES:
    t_srv, body = best(lambda: urllib.request.urlopen("http://127.0.0.1:%d/?%s" % (port, q), timeout=600).read())
    t_cli, out = best(lambda: subprocess.run([os.path.join(BIN, "bgt"), "view"] + va + [db], stdout=subprocess.PIPE, check=True).stdout, 3)
    line = "%-40s server %7.1f ms | bgt view (this repo) %7.1f ms" % (label, t_srv * 1e3, t_cli * 1e3)
    if os.path.exists(REF):
        t_ref, ro = best(lambda: subprocess.run([REF, "view"] + va + [db], stdout=subprocess.PIPE, check=True).stdout, 2)
        line += " | reference bgt view %8.1f ms (same bytes as this repo's: %s)" % (t_ref * 1e3, ro == out)
    print(line + " | %d lines" % body.count(b"\n"))
# throughput: many clients at once, each a stream of different small region queries (one thread per connection in the server,
# one pooled reader and HIP stream per query in flight)
import random
import threading
for n_clients in (1, 4, 16, 64, 256):
    per = 40
    bad = []

    def client(k):
        rnd = random.Random(k)
        for _ in range(per):
            a = 1000 + 10 * rnd.randrange(0, max(1, n_sites - 200))
            body = urllib.request.urlopen("http://127.0.0.1:%d/?C&r=11:%d-%d" % (port, a, a + 999), timeout=600).read()
            if body.count(b"\n") < 100:
                bad.append(a)
    th = [threading.Thread(target=client, args=(k,)) for k in range(n_clients)]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t
    print("%3d concurrent clients x %d region queries (100 sites, AC/AN): %7.0f queries/s, %d short answers" % (n_clients, per, n_clients * per / dt, len(bad)))
srv.terminate()
srv.wait()
if os.environ.get("SHOW_SERVER_LOG"):
    print("".join(open(os.path.join(tmp, "server.log")).readlines()[-60:]))
