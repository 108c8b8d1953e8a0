"""Message schema (interface.proto / interface.py) and socket framing
(utils.py) of the federated-learning example."""
