#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/probe/two_stream.py 2>&1 | tail -5
