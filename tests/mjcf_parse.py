"""Test infrastructure: a small reader for the MJCF assets the reference ships (gymnasium/envs/mujoco/assets/*.xml).

It resolves what MuJoCo's XML schema resolves before compilation -- built-in attribute defaults <- the file's top-level
<default> block <- the element's own attributes -- and returns the kinematic tree as plain Python data in document order, so
that `tests/test_mujoco_models_vs_xml.py` can compare the hand transcription in gymnasium_amd/envs/mujoco/models.py with the
XML field by field.  Only the standard library is used (xml.etree); nothing here is part of the product.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET

# MuJoCo's built-in defaults for the attributes that influence the physics (XML reference: body/joint, body/geom, option,
# compiler, actuator/motor).  `limited` has no value here: "auto" (autolimits) and the pre-2.3 "false" differ only for a joint
# that has a range but neither an own nor an inherited `limited` -- resolved_joint() reports such a joint as limited=None.
JOINT_BUILTIN = dict(type="hinge", pos=(0.0, 0.0, 0.0), axis=(0.0, 0.0, 1.0), armature=0.0, damping=0.0, stiffness=0.0, margin=0.0, ref=0.0,
                     springref=0.0, solreflimit=(0.02, 1.0), solimplimit=(0.9, 0.95, 0.001, 0.5, 2.0))
GEOM_BUILTIN = dict(type="sphere", contype=1, conaffinity=1, condim=3, density=1000.0, friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0,
                    solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.001, 0.5, 2.0), solmix=1.0)
OPTION_BUILTIN = dict(timestep=0.002, gravity=(0.0, 0.0, -9.81), integrator="Euler", solver="Newton", iterations=100, density=0.0, viscosity=0.0)
COMPILER_BUILTIN = dict(angle="degree", inertiafromgeom="auto", settotalmass=-1.0)


def _floats(s):
    return tuple(float(x) for x in s.split())


def _complete(value, builtin):
    """An attribute shorter than its full length overrides only the leading components (friction="0.9", solimp=".8 .8 .01")."""
    value = tuple(value)
    return value + tuple(builtin[len(value):])


def _bool(s):
    return {"true": True, "false": False}[s]


def _joint_attrs(el):
    out = {}
    for k, v in el.attrib.items():
        if k in ("armature", "damping", "stiffness", "margin", "ref", "springref"):
            out[k] = float(v)
        elif k in ("pos", "axis", "range"):
            out[k] = _floats(v)
        elif k in ("solreflimit", "solimplimit"):
            out[k] = _complete(_floats(v), JOINT_BUILTIN[k])
        elif k == "limited":
            out[k] = None if v == "auto" else _bool(v)
        elif k in ("type", "name"):
            out[k] = v
    return out


def _geom_attrs(el):
    out = {}
    for k, v in el.attrib.items():
        if k in ("contype", "conaffinity", "condim"):
            out[k] = int(v)
        elif k in ("density", "margin", "gap", "solmix"):
            out[k] = float(v)
        elif k in ("friction", "solref", "solimp"):
            out[k] = _complete(_floats(v), GEOM_BUILTIN[k])
        elif k in ("pos", "size", "fromto", "quat", "axisangle"):
            out[k] = _floats(v)
        elif k in ("type", "name"):
            out[k] = v
    return out


def _motor_attrs(el):
    out = {}
    for k, v in el.attrib.items():
        if k == "ctrlrange":
            out[k] = _floats(v)
        elif k == "ctrllimited":
            out[k] = None if v == "auto" else _bool(v)
        elif k == "gear":
            out[k] = _floats(v)[0]
        elif k in ("joint", "name"):
            out[k] = v
    return out


class Mjcf:
    """One parsed asset: compiler / option settings, the file's default block, world geoms, the body tree, actuators, tendons."""

    def __init__(self, path):
        root = ET.parse(path).getroot()
        assert root.tag == "mujoco"
        self.path = path
        comp = root.find("compiler")
        self.compiler = dict(COMPILER_BUILTIN)
        if comp is not None:
            for k, v in comp.attrib.items():
                if k == "settotalmass":
                    self.compiler[k] = float(v)
                elif k in ("angle", "inertiafromgeom", "coordinate"):
                    self.compiler[k] = v
        opt = root.find("option")
        self.option = dict(OPTION_BUILTIN)
        if opt is not None:
            for k, v in opt.attrib.items():
                if k in ("timestep", "density", "viscosity"):
                    self.option[k] = float(v)
                elif k == "gravity":
                    self.option[k] = _floats(v)
                elif k == "iterations":
                    self.option[k] = int(v)
                elif k in ("integrator", "solver"):
                    self.option[k] = v
        # only a flat top-level <default> occurs in the assets of the eleven v5 robots (no nested classes): assert it
        self.joint_default, self.geom_default, self.motor_default = {}, {}, {}
        defaults = root.findall("default")
        assert len(defaults) <= 1
        if defaults:
            assert defaults[0].find("default") is None, "nested default classes are not handled"
            for el in defaults[0]:
                if el.tag == "joint":
                    self.joint_default = _joint_attrs(el)
                elif el.tag == "geom":
                    self.geom_default = _geom_attrs(el)
                elif el.tag == "motor":
                    self.motor_default = _motor_attrs(el)
        assert not any("class" in el.attrib or "childclass" in el.attrib for el in root.iter()), "default classes are not handled"
        wb = root.find("worldbody")
        self.world_geoms = [self.resolved_geom(g) for g in wb.findall("geom")]
        self.bodies = [self._body(b) for b in wb.findall("body")]
        act = root.find("actuator")
        self.actuators = []
        for el in (act if act is not None else []):
            assert el.tag == "motor", el.tag
            a = dict(ctrllimited=None, ctrlrange=(0.0, 0.0), gear=1.0)
            a.update(self.motor_default)
            a.update(_motor_attrs(el))
            self.actuators.append(a)
        self.tendons = []
        ten = root.find("tendon")
        for el in (ten if ten is not None else []):
            assert el.tag == "fixed", el.tag
            # anything that would make a fixed tendon act on the dynamics
            assert not set(el.attrib) & {"limited", "range", "stiffness", "damping", "frictionloss", "springlength", "armature"}, el.attrib
            self.tendons.append((el.attrib["name"], tuple((j.attrib["joint"], float(j.attrib["coef"])) for j in el.findall("joint"))))
        self.tendon_actuated = any("tendon" in el.attrib for el in (act if act is not None else []))
        for tag in ("equality", "contact", "sensor", "keyframe"):
            assert root.find(tag) is None, f"<{tag}> is not handled"

    def resolved_joint(self, el):
        j = dict(JOINT_BUILTIN, limited=None, range=None, name=None)
        j.update(self.joint_default)
        j.update(_joint_attrs(el))
        return j

    def resolved_geom(self, el):
        g = dict(GEOM_BUILTIN, name=None, pos=(0.0, 0.0, 0.0), size=None, fromto=None, quat=None, axisangle=None)
        g.update(self.geom_default)
        g.update(_geom_attrs(el))
        return g

    def _body(self, el):
        assert el.find("inertial") is None, "explicit <inertial> is not handled (every asset uses inertiafromgeom)"
        return dict(name=el.attrib.get("name"), pos=_floats(el.attrib.get("pos", "0 0 0")),
                    quat=_floats(el.attrib["quat"]) if "quat" in el.attrib else None,
                    joints=[self.resolved_joint(j) for j in el.findall("joint")] + [self.resolved_joint(j) for j in el.findall("freejoint")],
                    geoms=[self.resolved_geom(g) for g in el.findall("geom")],
                    sites=[(s.attrib.get("name"), _floats(s.attrib.get("pos", "0 0 0"))) for s in el.findall("site")],
                    children=[self._body(b) for b in el.findall("body")])
