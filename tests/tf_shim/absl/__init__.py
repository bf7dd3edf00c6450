"""TEST-ONLY stand-in for absl (nlt/nlt_test.py defines its command-line flags at import; see ../README.md)."""
