"""Pinned-defaults wrapper reproducing /root/reference/part3/main.py on the new framework."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

from distributed_machine_learning_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main(part="part3")
