"""Score cache shared by runs (sqlite file or in memory) and a small HTTP
service in front of it, so that several workers -- or database sessions --
reuse training-free scores and training results.  Role: reference
src/query_api/ (score tables) and eva_service.py (HTTP endpoint)."""
import json
import sqlite3
import threading
from http.server import BaseHTTPRequestHandler, HTTPServer
from urllib.parse import parse_qs, urlparse


class ScoreCache(object):

    def __init__(self, path=":memory:"):
        self.db = sqlite3.connect(path, check_same_thread=False)
        self.lock = threading.Lock()
        with self.lock:
            self.db.execute(
                "CREATE TABLE IF NOT EXISTS scores (dataset TEXT, metric TEXT,"
                " arch TEXT, score REAL, PRIMARY KEY (dataset, metric, arch))")
            self.db.commit()

    def get(self, dataset, metric, arch):
        with self.lock:
            row = self.db.execute(
                "SELECT score FROM scores WHERE dataset=? AND metric=? AND "
                "arch=?", (dataset, metric, arch)).fetchone()
        return None if row is None else float(row[0])

    def put(self, dataset, metric, arch, score):
        with self.lock:
            self.db.execute("INSERT OR REPLACE INTO scores VALUES (?,?,?,?)",
                            (dataset, metric, arch, float(score)))
            self.db.commit()

    def top(self, dataset, metric, k):
        with self.lock:
            return self.db.execute(
                "SELECT arch, score FROM scores WHERE dataset=? AND metric=? "
                "ORDER BY score DESC LIMIT ?", (dataset, metric, k)).fetchall()

    def __len__(self):
        with self.lock:
            return self.db.execute("SELECT COUNT(*) FROM scores").fetchone()[0]


class ScoreService(object):
    """``GET /score?dataset=..&metric=..&arch=..`` -> {"score": s | null};
    ``POST /score`` with the same fields plus ``score`` stores one;
    ``GET /top?dataset=..&metric=..&k=..`` -> [[arch, score], ...]."""

    def __init__(self, cache, host="127.0.0.1", port=0):
        service = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def _reply(self, obj, code=200):
                body = json.dumps(obj).encode()
                self.send_response(code)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def do_GET(self):
                url = urlparse(self.path)
                q = {k: v[0] for k, v in parse_qs(url.query).items()}
                try:
                    if url.path == "/score":
                        self._reply({"score": service.cache.get(
                            q["dataset"], q["metric"], q["arch"])})
                    elif url.path == "/top":
                        self._reply(service.cache.top(
                            q["dataset"], q["metric"], int(q.get("k", 10))))
                    else:
                        self._reply({"error": "unknown path"}, 404)
                except KeyError as e:
                    self._reply({"error": "missing %s" % e}, 400)

            def do_POST(self):
                n = int(self.headers.get("Content-Length", 0))
                try:
                    q = json.loads(self.rfile.read(n) or b"{}")
                    service.cache.put(q["dataset"], q["metric"], q["arch"],
                                      q["score"])
                    self._reply({"ok": True})
                except (KeyError, ValueError) as e:
                    self._reply({"error": str(e)}, 400)

        self.cache = cache
        self.httpd = HTTPServer((host, port), Handler)
        self.port = self.httpd.server_address[1]
        self.thread = threading.Thread(target=self.httpd.serve_forever,
                                       daemon=True)

    def start(self):
        self.thread.start()
        return self

    def stop(self):
        self.httpd.shutdown()
        self.httpd.server_close()
