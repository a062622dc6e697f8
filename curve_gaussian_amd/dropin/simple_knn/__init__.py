"""PYTHONPATH shim for the reference's ``from simple_knn._C import distCUDA2``."""
