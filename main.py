"""Unified entry point: ``python main.py --master-ip HOST:PORT --rank R --num-nodes N
--sync {none,gather_scatter,all_reduce,ddp}`` (flags of /root/reference/part3/main.py:165-173
plus opt-in extras)."""
from distributed_machine_learning_amd.cli import main

if __name__ == "__main__":
    main()
