#!/bin/bash
# Start the prediction service in the background (the counterpart of the
# reference's examples/singa_easy/scripts/start.sh); arguments are passed on
# to serve.py.  The process id is kept in /tmp/singa_serve.pid for stop.sh.
cd "$(dirname "$0")/.."
nohup python serve.py "$@" > /tmp/singa_serve.log 2>&1 &
echo $! > /tmp/singa_serve.pid
echo "started (pid $(cat /tmp/singa_serve.pid)), log in /tmp/singa_serve.log"
