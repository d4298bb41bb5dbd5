#!/bin/bash
# ./tools/chol_rate_env.sh "<sizes>" "ENV=VAL ENV=VAL" ... : tools/chol_rate.py under several environments
sizes=$1; shift
for e in "$@"; do echo "== $e"; env $e timeout -k 10 400 python tools/chol_rate.py "$sizes" | cut -c1-75; done
