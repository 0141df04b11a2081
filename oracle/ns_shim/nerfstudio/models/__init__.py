"""Import shim (test infrastructure only) — see oracle/README.md."""
