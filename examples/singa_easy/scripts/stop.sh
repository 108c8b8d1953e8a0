#!/bin/bash
# Stop the prediction service started by start.sh (that process only).
if [ -f /tmp/singa_serve.pid ]; then
  kill "$(cat /tmp/singa_serve.pid)" 2>/dev/null && echo "stopped"
  rm -f /tmp/singa_serve.pid
else
  echo "no service recorded in /tmp/singa_serve.pid"
fi
