"""``WeightsExchange`` of interface.proto as a google.protobuf message class.

The class is built from a descriptor assembled here, so the example needs no
``protoc`` run and no generated ``interface_pb2.py`` (the reference checks one
in: examples/hfl/src/proto/interface_pb2.py); the bytes on the wire are those
``protoc``-generated code would produce for the same .proto file."""
from google.protobuf import descriptor_pb2, descriptor_pool

try:
    from google.protobuf.message_factory import GetMessageClass
except ImportError:                     # protobuf < 4.22
    from google.protobuf import message_factory
    GetMessageClass = message_factory.MessageFactory().GetPrototype

DEFAULT, SCATTER, GATHER = 0, 1, 2


def _build():
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(
        name="singa_amd_hfl_interface.proto", package="interface",
        syntax="proto3")
    op = fd.enum_type.add(name="Op")
    for name, number in (("DEFAULT", 0), ("SCATTER", 1), ("GATHER", 2)):
        op.value.add(name=name, number=number)
    msg = fd.message_type.add(name="WeightsExchange")
    msg.field.add(name="op_type", number=1, label=F.LABEL_OPTIONAL,
                  type=F.TYPE_ENUM, type_name=".interface.Op")
    # map<string, bytes> is a repeated nested entry message (key = 1,
    # value = 2) flagged map_entry
    entry = msg.nested_type.add(name="WeightsEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, label=F.LABEL_OPTIONAL,
                    type=F.TYPE_STRING)
    entry.field.add(name="value", number=2, label=F.LABEL_OPTIONAL,
                    type=F.TYPE_BYTES)
    msg.field.add(name="weights", number=2, label=F.LABEL_REPEATED,
                  type=F.TYPE_MESSAGE,
                  type_name=".interface.WeightsExchange.WeightsEntry")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return GetMessageClass(
        pool.FindMessageTypeByName("interface.WeightsExchange"))


WeightsExchange = _build()
