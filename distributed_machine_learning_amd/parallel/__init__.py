from .comm import Communicator, HostCommunicator, RcclCommunicator, init_communicator  # noqa: F401
from .flat import FlatModule, plan_buckets, plan_slots  # noqa: F401
from .strategies import (GradSync, NoSync, GatherScatterSync, AllReduceSync,  # noqa: F401
                         BucketedSync)
