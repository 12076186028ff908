from .comm import Communicator, RcclCommunicator, init_communicator  # noqa: F401
