"""Puts the product package dir (`neural-backed-decision-trees_amd/`, not an importable name) on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "neural-backed-decision-trees_amd")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def add(oracle=False):
    if PKG_DIR not in sys.path:
        sys.path.insert(0, PKG_DIR)
    if oracle and ORACLE_DIR not in sys.path:
        sys.path.insert(0, ORACLE_DIR)
