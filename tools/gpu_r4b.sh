#!/bin/bash
set -u
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
