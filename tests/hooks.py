"""Is this process one that asked libhrfd for its behaviour-changing test hooks?  (tests/conftest.py sets
HRFD_DEBUG_HOOKS=1 unless HRFD_HOOKS_OFF=1: the shipped state, run by tests/test_gpu_hooks_off.py in a child process.)"""
import os

HOOKS_ON = os.environ.get("HRFD_DEBUG_HOOKS") == "1"
