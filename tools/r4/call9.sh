#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/scan_alone.py 32 2>&1 | tail -15
