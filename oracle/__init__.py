"""CPU oracle for the AlphaSnake-Zero self-play hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import, link or execute it, and then
only as the checker / the CPU baseline, never as the thing measured or shipped.
The product path (``alphasnake-zero_amd/``) fails loudly when its HIP library
is missing; it never falls back to this code.
"""
