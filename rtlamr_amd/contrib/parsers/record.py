"""Message.Record / Message.String of the reference's message types -- the csv.Recorder half of protocol.Message
(parse.go:78-84) that main.go's CSV / JSON / plain encoders print.

Optional and off the hot path: `protocol.Message.Record()` and `str(message)` import this module on first use.  The
columns and formats follow scm/scm.go:139-154, scmplus/scmplus.go:129-150, idm/idm.go:176-221,
netidm/netidm.go:186-235 and r900/r900.go:278-302; messages of a type this module does not know fall back to the
four identifying methods.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple


def _gohex(v: int) -> str:
    """"0x" + strconv.FormatUint(v, 16): lower case, no padding."""
    return "0x%x" % v


def _hx(v: int, w: int) -> str:
    """fmt's 0x%0*X."""
    return "0x%0*X" % (w, v)


def _raw(b) -> str:
    """fmt's %02X of a byte slice: its hex digits, upper case."""
    return bytes(b).hex().upper()


def _scm(m) -> Tuple[List[str], str]:
    rec = [str(m.ID), str(m.Type), _gohex(m.TamperPhy), _gohex(m.TamperEnc), str(m.Consumption), _gohex(m.ChecksumVal)]
    txt = "{ID:%8d Type:%2d Tamper:{Phy:%02X Enc:%02X} Consumption:%8d CRC:0x%04X}" % (
        m.ID, m.Type, m.TamperPhy, m.TamperEnc, m.Consumption, m.ChecksumVal)
    return rec, txt


def _scmplus(m) -> Tuple[List[str], str]:
    rec = [_gohex(m.FrameSync), _gohex(m.ProtocolID), _gohex(m.EndpointType), str(m.EndpointID), str(m.Consumption),
           _gohex(m.Tamper), _gohex(m.PacketCRC)]
    txt = "{ProtocolID:0x%02X EndpointType:0x%02X EndpointID:%10d Consumption:%10d Tamper:0x%04X PacketCRC:0x%04X}" % (
        m.ProtocolID, m.EndpointType, m.EndpointID, m.Consumption, m.Tamper, m.PacketCRC)
    return rec, txt


def _r900(m) -> Tuple[List[str], str]:
    rec = [str(v) for v in (m.ID, m.Unkn1, m.NoUse, m.BackFlow, m.Consumption, m.Unkn3, m.Leak, m.LeakNow)]
    txt = "{ID:%10d Unkn1:0x%02X NoUse:%2d BackFlow:%1d Consumption:%8d Unkn3:0x%02X Leak:%2d LeakNow:%1d}" % (
        m.ID, m.Unkn1, m.NoUse, m.BackFlow, m.Consumption, m.Unkn3, m.Leak, m.LeakNow)
    return rec, txt


def _columns(fields) -> Tuple[List[str], str]:
    """fields: (name, CSV form or list of CSV forms, plain-text form or None = the CSV form)"""
    rec: List[str] = []
    for _, csv, _ in fields:
        rec.extend(csv if isinstance(csv, list) else [csv])
    txt = "{" + " ".join(f"{n}:{(t if t is not None else c)}" for n, c, t in fields) + "}"
    return rec, txt


def _idm(m) -> Tuple[List[str], str]:
    iv = m.DifferentialConsumptionIntervals
    return _columns([
        ("Preamble", _hx(m.Preamble, 8), None), ("PacketTypeID", _hx(m.PacketTypeID, 2), None),
        ("PacketLength", _hx(m.PacketLength, 2), None), ("HammingCode", _hx(m.HammingCode, 2), None),
        ("ApplicationVersion", _hx(m.ApplicationVersion, 2), None), ("ERTType", _hx(m.ERTType, 2), None),
        ("ERTSerialNumber", str(m.ERTSerialNumber), "% 10d" % m.ERTSerialNumber),
        ("ConsumptionIntervalCount", str(m.ConsumptionIntervalCount), None),
        ("ModuleProgrammingState", _hx(m.ModuleProgrammingState, 2), None),
        ("TamperCounters", _raw(m.TamperCounters), None),
        ("AsynchronousCounters", _hx(m.AsynchronousCounters, 2), None),
        ("PowerOutageFlags", _raw(m.PowerOutageFlags), None),
        ("LastConsumptionCount", str(m.LastConsumptionCount), None),
        ("DifferentialConsumptionIntervals", [str(v) for v in iv], "[" + " ".join(str(v) for v in iv) + "]"),
        ("TransmitTimeOffset", str(m.TransmitTimeOffset), None),
        ("SerialNumberCRC", _hx(m.SerialNumberCRC, 4), None), ("PacketCRC", _hx(m.PacketCRC, 4), None)])


def _netidm(m) -> Tuple[List[str], str]:
    iv = m.DifferentialConsumptionIntervals
    return _columns([
        ("Preamble", _hx(m.Preamble, 8), None), ("ProtocolID", _hx(m.ProtocolID, 2), None),
        ("PacketLength", _hx(m.PacketLength, 2), None), ("HammingCode", _hx(m.HammingCode, 2), None),
        ("ApplicationVersion", _hx(m.ApplicationVersion, 2), None), ("ERTType", _hx(m.ERTType, 2), None),
        ("ERTSerialNumber", str(m.ERTSerialNumber), "% 10d" % m.ERTSerialNumber),
        ("ConsumptionIntervalCount", str(m.ConsumptionIntervalCount), None),
        ("ProgrammingState", _hx(m.ProgrammingState, 2), None),
        ("LastGeneration", str(m.LastGeneration), None), ("LastConsumption", str(m.LastConsumption), None),
        ("LastConsumptionNet", str(m.LastConsumptionNet), None),
        ("DifferentialConsumptionIntervals", [str(v) for v in iv], "[" + " ".join(str(v) for v in iv) + "]"),
        ("TransmitTimeOffset", str(m.TransmitTimeOffset), None),
        ("SerialNumberCRC", _hx(m.SerialNumberCRC, 4), None), ("PacketCRC", _hx(m.PacketCRC, 4), None)])


# R900BCD embeds r900.R900 (r900bcd/r900bcd.go:39-41): Record() and String() are R900's, promoted
_BY_TYPE: Dict[str, Callable] = {"SCM": _scm, "SCM+": _scmplus, "IDM": _idm, "NetIDM": _netidm, "R900": _r900,
                                 "R900BCD": _r900}


def _generic(m) -> Tuple[List[str], str]:
    ck = bytes(m.Checksum()).hex().upper()
    return ([m.MsgType(), str(m.MeterID()), str(m.MeterType()), ck],
            f"{{{m.MsgType()} ID:{m.MeterID()} Type:{m.MeterType()} Checksum:0x{ck}}}")


def record(m) -> List[str]:
    """Message.Record(): the CSV columns."""
    return _BY_TYPE.get(m.MsgType(), _generic)(m)[0]


def string(m) -> str:
    """Message.String(): the plain-text form."""
    return _BY_TYPE.get(m.MsgType(), _generic)(m)[1]
