"""TEST INFRASTRUCTURE ONLY: CPU oracle for the hash-groupby / hash-join path (see oracle.h)."""
