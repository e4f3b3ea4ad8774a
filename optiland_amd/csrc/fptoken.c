/* _fptoken -- native inner loop of optiland_amd/fingerprint.py (host runtime, CPython API).
 *
 * The change detector of the drop-in walks ~400 attribute values of the live reference
 * objects on EVERY trace; in pure Python that walk is the largest host cost of a small
 * trace (~90 us of a ~250 us call on the GPU box's EPYC).  This module implements exactly
 * the walk -- `tok`, `dict_tokens`, `obj` and `surface_token` of fingerprint.py -- with the same
 * results as their Python definitions (tests/test_fingerprint.py compares the two token
 * trees on every lens of optiland.samples and replays the mutation suite on both).
 *
 *   float / int / str / bool / None / complex -> the value itself
 *   torch.Tensor (and subclasses)             -> ("T", id, _version); kept alive
 *   numpy.ndarray                             -> ("A", shape, bytes)   size <= big
 *                                                ("A", id, shape)      otherwise; kept alive
 *   numpy scalar                              -> .item()
 *   list / tuple                              -> tuple of element tokens
 *   dict                                      -> None   (the reference keeps caches there)
 *   anything else                             -> ("O", type name, id); kept alive
 *
 * No device access, no torch / numpy headers: the two array types are handed over once by
 * `configure(tensor_type, ndarray_type, generic_type, big)`.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static PyObject *g_tensor = NULL, *g_ndarray = NULL, *g_generic = NULL;
/* Python callables for the two rare cases (fingerprint.py): a tensor that requires grad
 * (writes through `.data` leave `_version` alone: never trusted), a numpy array too big to
 * carry by value (content digest) */
static PyObject *g_grad_tok = NULL, *g_big_tok = NULL;
static PyObject* s_requires_grad = NULL;
static Py_ssize_t g_big = 1 << 14;
static PyObject *s_T, *s_A, *s_O, *s_version, *s_shape, *s_size, *s_tobytes, *s_item, *s_dict,
    *s_reference_cs, *s_propagation_model, *s_a, *s_b, *s_jones, *s__jones, *s_geometry,
    *s_interaction_model, *s_cs, *s_zernike, *s_material_pre, *s_material_post, *s_aperture,
    *s_coating, *s_thickness, *s_is_stop;

static PyObject* id_of(PyObject* v) { return PyLong_FromVoidPtr((void*)v); }

static PyObject* tok(PyObject* v, PyObject* keep);

static PyObject* tensor_tok(PyObject* v, PyObject* keep) {
  if (PyList_Append(keep, v) < 0) return NULL;
  if (g_grad_tok) {
    PyObject* rg = PyObject_GetAttr(v, s_requires_grad);
    if (!rg) return NULL;
    const int yes = PyObject_IsTrue(rg);
    Py_DECREF(rg);
    if (yes < 0) return NULL;
    if (yes) return PyObject_CallOneArg(g_grad_tok, v);
  }
  PyObject* ver = PyObject_GetAttr(v, s_version);
  if (!ver) return NULL;
  PyObject* idv = id_of(v);
  if (!idv) { Py_DECREF(ver); return NULL; }
  PyObject* out = PyTuple_Pack(3, s_T, idv, ver);
  Py_DECREF(idv);
  Py_DECREF(ver);
  return out;
}

static PyObject* ndarray_tok(PyObject* v, PyObject* keep) {
  PyObject* size = PyObject_GetAttr(v, s_size);
  if (!size) return NULL;
  const Py_ssize_t n = PyLong_AsSsize_t(size);
  Py_DECREF(size);
  if (n == -1 && PyErr_Occurred()) return NULL;
  PyObject* shape = PyObject_GetAttr(v, s_shape);
  if (!shape) return NULL;
  PyObject* out = NULL;
  if (n <= g_big) {
    PyObject* bytes = PyObject_CallMethodNoArgs(v, s_tobytes);
    if (bytes) {
      out = PyTuple_Pack(3, s_A, shape, bytes);
      Py_DECREF(bytes);
    }
  } else if (PyList_Append(keep, v) == 0) {
    if (g_big_tok) {
      out = PyObject_CallOneArg(g_big_tok, v);
    } else {
      PyObject* idv = id_of(v);
      if (idv) {
        out = PyTuple_Pack(3, s_A, idv, shape);
        Py_DECREF(idv);
      }
    }
  }
  Py_DECREF(shape);
  return out;
}

static PyObject* seq_tok(PyObject* v, PyObject* keep) {
  PyObject* fast = PySequence_Fast(v, "sequence");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  PyObject* out = PyTuple_New(n);
  if (!out) { Py_DECREF(fast); return NULL; }
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* t = tok(PySequence_Fast_GET_ITEM(fast, i), keep);
    if (!t) { Py_DECREF(out); Py_DECREF(fast); return NULL; }
    PyTuple_SET_ITEM(out, i, t);
  }
  Py_DECREF(fast);
  return out;
}

static PyObject* tok(PyObject* v, PyObject* keep) {
  PyTypeObject* t = Py_TYPE(v);
  if (t == &PyFloat_Type || t == &PyLong_Type || t == &PyUnicode_Type || t == &PyBool_Type ||
      v == Py_None || t == &PyComplex_Type) {
    Py_INCREF(v);
    return v;
  }
  if ((PyObject*)t == g_tensor) return tensor_tok(v, keep);
  if (g_tensor) {
    const int r = PyObject_IsInstance(v, g_tensor);  /* Parameter and other subclasses */
    if (r < 0) return NULL;
    if (r) return tensor_tok(v, keep);
  }
  if (g_ndarray) {
    const int r = (PyObject*)t == g_ndarray ? 1 : PyObject_IsInstance(v, g_ndarray);
    if (r < 0) return NULL;
    if (r) return ndarray_tok(v, keep);
  }
  if (g_generic) {
    const int r = PyObject_IsInstance(v, g_generic);
    if (r < 0) return NULL;
    if (r) return PyObject_CallMethodNoArgs(v, s_item);
  }
  if (t == &PyList_Type || t == &PyTuple_Type) return seq_tok(v, keep);
  if (t == &PyDict_Type) Py_RETURN_NONE;
  if (PyList_Append(keep, v) < 0) return NULL;
  PyObject* name = PyUnicode_FromString(t->tp_name);
  if (!name) return NULL;
  /* t.__name__ is the last component of tp_name for heap and static types alike */
  const char* dot = strrchr(t->tp_name, '.');
  if (dot) {
    Py_DECREF(name);
    name = PyUnicode_FromString(dot + 1);
    if (!name) return NULL;
  }
  PyObject* idv = id_of(v);
  if (!idv) { Py_DECREF(name); return NULL; }
  PyObject* out = PyTuple_Pack(3, s_O, name, idv);
  Py_DECREF(name);
  Py_DECREF(idv);
  return out;
}

static PyObject* py_tok(PyObject* self, PyObject* args) {
  PyObject *v, *keep;
  if (!PyArg_ParseTuple(args, "OO!", &v, &PyList_Type, &keep)) return NULL;
  return tok(v, keep);
}

static PyObject* dict_tokens(PyObject* d, PyObject* skip, PyObject* keep) {
  PyObject* out = PyList_New(0);
  if (!out) return NULL;
  PyObject *k, *v;
  Py_ssize_t pos = 0;
  while (PyDict_Next(d, &pos, &k, &v)) {
    if (skip != Py_None) {
      const int in = PySequence_Contains(skip, k);
      if (in < 0) { Py_DECREF(out); return NULL; }
      if (in) continue;
    }
    PyObject* t = tok(v, keep);
    if (!t) { Py_DECREF(out); return NULL; }
    PyObject* item = t;
    if (skip != Py_None) {
      item = PyTuple_Pack(2, k, t);
      Py_DECREF(t);
      if (!item) { Py_DECREF(out); return NULL; }
    }
    const int rc = PyList_Append(out, item);
    Py_DECREF(item);
    if (rc < 0) { Py_DECREF(out); return NULL; }
  }
  return out;
}

static PyObject* type_name(PyObject* o) {
  const char* tn = Py_TYPE(o)->tp_name;
  const char* dot = strrchr(tn, '.');
  return PyUnicode_FromString(dot ? dot + 1 : tn);
}

/* getattr(o, name, None): new reference, AttributeError -> None */
static PyObject* attr_or_none(PyObject* o, PyObject* name) {
  PyObject* v = PyObject_GetAttr(o, name);
  if (!v) {
    if (!PyErr_ExceptionMatches(PyExc_AttributeError)) return NULL;
    PyErr_Clear();
    Py_RETURN_NONE;
  }
  return v;
}

/* fingerprint._obj -- one level of o.__dict__, memoised by id(o) */
static PyObject* obj(PyObject* o, PyObject* keep, PyObject* memo, PyObject* skip) {
  if (o == Py_None) Py_RETURN_NONE;
  PyObject* key = id_of(o);
  if (!key) return NULL;
  PyObject* hit = PyDict_GetItemWithError(memo, key);  /* borrowed */
  if (hit) {
    Py_DECREF(key);
    Py_INCREF(hit);
    return hit;
  }
  if (PyErr_Occurred()) { Py_DECREF(key); return NULL; }
  if (PyList_Append(keep, o) < 0) { Py_DECREF(key); return NULL; }
  PyObject* out = NULL;
  PyObject* d = NULL;
#if PY_VERSION_HEX < 0x030C0000
  /* fast path while the interpreter still keeps instance dicts behind a plain pointer
     (CPython < 3.12); everything newer takes the public attribute lookup below */
  PyObject** dictptr = _PyObject_GetDictPtr(o);
  if (dictptr && *dictptr && PyDict_Check(*dictptr)) {
    d = *dictptr;
    Py_INCREF(d);
  }
#endif
  if (!d) {
    d = PyObject_GetAttr(o, s_dict);
    if (!d) PyErr_Clear();
  }
  if (d && PyDict_Check(d)) {
    PyObject* toks = dict_tokens(d, skip, keep);
    if (toks) {
      PyObject* name = type_name(o);
      if (name) {
        out = PyTuple_Pack(3, name, key, toks);
        Py_DECREF(name);
      }
      Py_DECREF(toks);
    }
  } else {
    out = tok(o, keep);  /* no __dict__ (slots, builtins): by value / identity */
  }
  Py_XDECREF(d);
  if (out && PyDict_SetItem(memo, key, out) < 0) {
    Py_DECREF(out);
    out = NULL;
  }
  Py_DECREF(key);
  return out;
}

static PyObject* py_obj(PyObject* self, PyObject* args) {
  PyObject *o, *keep, *memo, *skip = Py_None;
  if (!PyArg_ParseTuple(args, "OO!O!|O", &o, &PyList_Type, &keep, &PyDict_Type, &memo, &skip))
    return NULL;
  return obj(o, keep, memo, skip);
}

/* fingerprint._cs: the chain cs, cs.reference_cs, ... */
static PyObject* cs_chain(PyObject* cs, PyObject* keep, PyObject* memo) {
  PyObject* list = PyList_New(0);
  if (!list) return NULL;
  Py_INCREF(cs);
  while (cs != Py_None) {
    PyObject* t = obj(cs, keep, memo, Py_None);
    if (!t || PyList_Append(list, t) < 0) { Py_XDECREF(t); Py_DECREF(cs); Py_DECREF(list); return NULL; }
    Py_DECREF(t);
    PyObject* next = attr_or_none(cs, s_reference_cs);
    Py_DECREF(cs);
    if (!next) { Py_DECREF(list); return NULL; }
    cs = next;
  }
  Py_DECREF(cs);
  PyObject* out = PyList_AsTuple(list);
  Py_DECREF(list);
  return out;
}

/* fingerprint._material: (obj(m), type(m.propagation_model).__name__) */
static PyObject* material(PyObject* m, PyObject* keep, PyObject* memo) {
  if (m == Py_None) Py_RETURN_NONE;
  PyObject* a = obj(m, keep, memo, Py_None);
  if (!a) return NULL;
  PyObject* pm = attr_or_none(m, s_propagation_model);
  if (!pm) { Py_DECREF(a); return NULL; }
  PyObject* name = type_name(pm);
  Py_DECREF(pm);
  if (!name) { Py_DECREF(a); return NULL; }
  PyObject* out = PyTuple_Pack(2, a, name);
  Py_DECREF(a);
  Py_DECREF(name);
  return out;
}

/* fingerprint._aperture: (obj(ap), tuple(_aperture(c) for c in (ap.a, ap.b) if it is an
 * object with a __dict__)) */
static PyObject* aperture(PyObject* ap, PyObject* keep, PyObject* memo) {
  if (ap == Py_None) Py_RETURN_NONE;
  PyObject* sub = PyList_New(0);
  if (!sub) return NULL;
  PyObject* names[2] = {s_a, s_b};
  for (int i = 0; i < 2; ++i) {
    PyObject* c = attr_or_none(ap, names[i]);
    if (!c) { Py_DECREF(sub); return NULL; }
    if (c != Py_None) {
      const int has = PyObject_HasAttr(c, s_dict);
      if (has) {
        PyObject* t = aperture(c, keep, memo);
        if (!t || PyList_Append(sub, t) < 0) { Py_XDECREF(t); Py_DECREF(c); Py_DECREF(sub); return NULL; }
        Py_DECREF(t);
      }
    }
    Py_DECREF(c);
  }
  PyObject* self_tok = obj(ap, keep, memo, Py_None);
  if (!self_tok) { Py_DECREF(sub); return NULL; }
  PyObject* subt = PyList_AsTuple(sub);
  Py_DECREF(sub);
  if (!subt) { Py_DECREF(self_tok); return NULL; }
  PyObject* out = PyTuple_Pack(2, self_tok, subt);
  Py_DECREF(self_tok);
  Py_DECREF(subt);
  return out;
}

/* fingerprint._coating: (obj(c), obj(c.jones or c._jones)) */
static PyObject* coating(PyObject* c, PyObject* keep, PyObject* memo) {
  if (c == Py_None) Py_RETURN_NONE;
  PyObject* a = obj(c, keep, memo, Py_None);
  if (!a) return NULL;
  PyObject* j = attr_or_none(c, s_jones);
  if (!j) { Py_DECREF(a); return NULL; }
  const int truth = PyObject_IsTrue(j);
  if (truth < 0) { Py_DECREF(a); Py_DECREF(j); return NULL; }
  if (!truth) {
    Py_DECREF(j);
    j = attr_or_none(c, s__jones);
    if (!j) { Py_DECREF(a); return NULL; }
  }
  PyObject* jt = obj(j, keep, memo, Py_None);
  Py_DECREF(j);
  if (!jt) { Py_DECREF(a); return NULL; }
  PyObject* out = PyTuple_Pack(2, a, jt);
  Py_DECREF(a);
  Py_DECREF(jt);
  return out;
}

/* fingerprint.surface_token(s, keep, memo, surface_skip) */
static PyObject* py_surface_token(PyObject* self, PyObject* args) {
  PyObject *s, *keep, *memo, *skip;
  if (!PyArg_ParseTuple(args, "OO!O!O", &s, &PyList_Type, &keep, &PyDict_Type, &memo, &skip))
    return NULL;
  PyObject* item[12] = {0};
  PyObject *geom = NULL, *im = NULL, *tmp = NULL, *out = NULL;
  geom = PyObject_GetAttr(s, s_geometry);
  if (!geom) goto done;
  im = attr_or_none(s, s_interaction_model);
  if (!im) goto done;
  if (!(item[0] = type_name(s))) goto done;
  if (!(item[1] = id_of(s))) goto done;
  if (!(item[2] = obj(geom, keep, memo, Py_None))) goto done;
  if (!(tmp = attr_or_none(geom, s_cs))) goto done;
  item[3] = cs_chain(tmp, keep, memo);
  Py_CLEAR(tmp);
  if (!item[3]) goto done;
  if (!(tmp = attr_or_none(geom, s_zernike))) goto done;
  item[4] = obj(tmp, keep, memo, Py_None);
  Py_CLEAR(tmp);
  if (!item[4]) goto done;
  if (!(tmp = attr_or_none(s, s_material_pre))) goto done;
  item[5] = material(tmp, keep, memo);
  Py_CLEAR(tmp);
  if (!item[5]) goto done;
  if (!(tmp = attr_or_none(s, s_material_post))) goto done;
  item[6] = material(tmp, keep, memo);
  Py_CLEAR(tmp);
  if (!item[6]) goto done;
  if (!(tmp = attr_or_none(s, s_aperture))) goto done;
  item[7] = aperture(tmp, keep, memo);
  Py_CLEAR(tmp);
  if (!item[7]) goto done;
  if (!(item[8] = obj(im, keep, memo, skip))) goto done;
  if (im == Py_None) {
    item[9] = Py_None;
    Py_INCREF(Py_None);
  } else {
    if (!(tmp = attr_or_none(im, s_coating))) goto done;
    item[9] = coating(tmp, keep, memo);
    Py_CLEAR(tmp);
    if (!item[9]) goto done;
  }
  if (!(tmp = attr_or_none(s, s_thickness))) goto done;
  item[10] = tok(tmp, keep);
  Py_CLEAR(tmp);
  if (!item[10]) goto done;
  if (!(tmp = PyObject_GetAttr(s, s_is_stop))) {
    if (!PyErr_ExceptionMatches(PyExc_AttributeError)) goto done;
    PyErr_Clear();
    item[11] = Py_False;
    Py_INCREF(Py_False);
  } else {
    const int t = PyObject_IsTrue(tmp);
    Py_CLEAR(tmp);
    if (t < 0) goto done;
    item[11] = t ? Py_True : Py_False;
    Py_INCREF(item[11]);
  }
  out = PyTuple_New(12);
  if (out)
    for (int i = 0; i < 12; ++i) {
      PyTuple_SET_ITEM(out, i, item[i]);
      item[i] = NULL;
    }
done:
  for (int i = 0; i < 12; ++i) Py_XDECREF(item[i]);
  Py_XDECREF(geom);
  Py_XDECREF(im);
  Py_XDECREF(tmp);
  return out;
}

/* dict_tokens(d, skip, keep): [tok(v) for v in d.values()] when skip is None, else
 * [(k, tok(v)) for k, v in d.items() if k not in skip] */
static PyObject* py_dict_tokens(PyObject* self, PyObject* args) {
  PyObject *d, *skip, *keep;
  if (!PyArg_ParseTuple(args, "O!OO!", &PyDict_Type, &d, &skip, &PyList_Type, &keep)) return NULL;
  return dict_tokens(d, skip, keep);
}

static PyObject* py_configure(PyObject* self, PyObject* args) {
  PyObject *tensor, *ndarray, *generic, *grad_tok = NULL, *big_tok = NULL;
  Py_ssize_t big;
  if (!PyArg_ParseTuple(args, "OOOn|OO", &tensor, &ndarray, &generic, &big, &grad_tok, &big_tok))
    return NULL;
  Py_XDECREF(g_tensor); Py_XDECREF(g_ndarray); Py_XDECREF(g_generic);
  Py_XDECREF(g_grad_tok); Py_XDECREF(g_big_tok);
  Py_INCREF(tensor); Py_INCREF(ndarray); Py_INCREF(generic);
  g_tensor = tensor; g_ndarray = ndarray; g_generic = generic;
  g_grad_tok = (grad_tok && grad_tok != Py_None) ? (Py_INCREF(grad_tok), grad_tok) : NULL;
  g_big_tok = (big_tok && big_tok != Py_None) ? (Py_INCREF(big_tok), big_tok) : NULL;
  g_big = big;
  Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"tok", py_tok, METH_VARARGS, "token of one attribute value"},
    {"dict_tokens", py_dict_tokens, METH_VARARGS, "tokens of the values of an object's __dict__"},
    {"obj", py_obj, METH_VARARGS, "obj(o, keep, memo[, skip]): one memoised level of o.__dict__"},
    {"surface_token", py_surface_token, METH_VARARGS,
     "surface_token(surface, keep, memo, skip): everything pack_surfaces reads of one Surface"},
    {"configure", py_configure, METH_VARARGS, "configure(tensor_type, ndarray_type, generic_type, big)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fptoken",
                                    "native inner loop of optiland_amd.fingerprint", -1, methods};

PyMODINIT_FUNC PyInit__fptoken(void) {
  s_T = PyUnicode_InternFromString("T");
  s_A = PyUnicode_InternFromString("A");
  s_O = PyUnicode_InternFromString("O");
  s_version = PyUnicode_InternFromString("_version");
  s_requires_grad = PyUnicode_InternFromString("requires_grad");
  s_shape = PyUnicode_InternFromString("shape");
  s_size = PyUnicode_InternFromString("size");
  s_tobytes = PyUnicode_InternFromString("tobytes");
  s_item = PyUnicode_InternFromString("item");
  s_dict = PyUnicode_InternFromString("__dict__");
  s_reference_cs = PyUnicode_InternFromString("reference_cs");
  s_propagation_model = PyUnicode_InternFromString("propagation_model");
  s_a = PyUnicode_InternFromString("a");
  s_b = PyUnicode_InternFromString("b");
  s_jones = PyUnicode_InternFromString("jones");
  s__jones = PyUnicode_InternFromString("_jones");
  s_geometry = PyUnicode_InternFromString("geometry");
  s_interaction_model = PyUnicode_InternFromString("interaction_model");
  s_cs = PyUnicode_InternFromString("cs");
  s_zernike = PyUnicode_InternFromString("zernike");
  s_material_pre = PyUnicode_InternFromString("material_pre");
  s_material_post = PyUnicode_InternFromString("material_post");
  s_aperture = PyUnicode_InternFromString("aperture");
  s_coating = PyUnicode_InternFromString("coating");
  s_thickness = PyUnicode_InternFromString("thickness");
  s_is_stop = PyUnicode_InternFromString("is_stop");
  return PyModule_Create(&moddef);
}
