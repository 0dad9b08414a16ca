#!/bin/bash
# A/B aid: runs tools/rl_check.py under each "NAME=ENV..." variant given on the command line
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name ($envs)"
  env $envs python tools/rl_check.py 2>&1 | grep median
done
