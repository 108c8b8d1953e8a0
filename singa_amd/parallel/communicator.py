"""placeholder, replaced by the communicator milestone"""


class NcclIdHolder(object):
    pass


class Communicator(object):
    pass
