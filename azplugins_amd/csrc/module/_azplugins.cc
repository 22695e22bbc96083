// _azplugins.cc -- HOOMD-free pybind11 module with the class names the reference
// registers in hoomd.azplugins._azplugins (src/module.cc:110-166 through the
// export_*.cc.inc templates) for the force-compute path, and the C++ half of their
// parameter interface: setParams(type..., dict) / getParams(type...) -> dict through the
// same dict constructors and asDict() / toPython() arithmetic as the reference's
// param_type structs (file:line at each codec), implemented once in libazp
// (azp_*_params_make / _unpack, csrc/azp_host.cpp) and only marshalled here.
//
// What a class of this module is: the host-side table keeper of one potential -- per
// type pair (pairs) or per bond type (bonds) the packed param_type struct, r_cut, r_on and
// the shift mode -- i.e. the state HOOMD's PotentialPair<E> / PotentialBond<E> hold on the
// host and hand to the kernel driver as device tables. The constructor takes the type
// names (HOOMD: a SystemDefinition, absent here); the "GPU" classes are the same type
// with on_gpu = true. azplugins_amd.pair / .bond keep one of these per potential and
// upload params_bytes(), rcutsq(), ronsq() for libazp's kernels, so the dict -> struct ->
// dict round trip the reference asserts with == (src/pytest/test_pair.py:349) runs in C++.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/azp.h"

namespace py = pybind11;

namespace
{
// ---- dict <-> param_type codecs ----
template<class P> struct Codec;

template<> struct Codec<azp_plj_params>
    { // src/PairEvaluatorPerturbedLennardJones.h:33-54
    static azp_plj_params from_dict(const py::dict& v)
        {
        azp_plj_params p;
        azp_plj_params_make(v["epsilon"].cast<double>(), v["sigma"].cast<double>(), v["attraction_scale_factor"].cast<double>(), &p);
        return p;
        }
    static py::dict to_dict(const azp_plj_params& p)
        {
        double e, s, l;
        azp_plj_params_unpack(&p, &e, &s, &l);
        py::dict v;
        v["sigma"] = s; v["epsilon"] = e; v["attraction_scale_factor"] = l;
        return v;
        }
    };
template<> struct Codec<azp_hertz_params>
    { // src/PairEvaluatorHertz.h:23-47
    static azp_hertz_params from_dict(const py::dict& v) { return azp_hertz_params {v["epsilon"].cast<double>()}; }
    static py::dict to_dict(const azp_hertz_params& p)
        {
        py::dict v;
        v["epsilon"] = p.epsilon;
        return v;
        }
    };
template<> struct Codec<azp_yukawa_params>
    { // src/PairEvaluatorExpandedYukawa.h:23-53
    static azp_yukawa_params from_dict(const py::dict& v)
        {
        return azp_yukawa_params {v["epsilon"].cast<double>(), v["kappa"].cast<double>(), v["delta"].cast<double>(), 0.0};
        }
    static py::dict to_dict(const azp_yukawa_params& p)
        {
        py::dict v;
        v["epsilon"] = p.epsilon; v["kappa"] = p.kappa; v["delta"] = p.delta;
        return v;
        }
    };
template<> struct Codec<azp_colloid_params>
    { // src/PairEvaluatorColloid.h:23-57
    static azp_colloid_params from_dict(const py::dict& v)
        {
        azp_colloid_params p;
        azp_colloid_params_make(v["A"].cast<double>(), v["a_1"].cast<double>(), v["a_2"].cast<double>(), v["sigma"].cast<double>(), &p);
        return p;
        }
    static py::dict to_dict(const azp_colloid_params& p)
        {
        double A, a1, a2, s;
        azp_colloid_params_unpack(&p, &A, &a1, &a2, &s);
        py::dict v;
        v["A"] = A; v["a_1"] = a1; v["a_2"] = a2; v["sigma"] = s;
        return v;
        }
    };
template<> struct Codec<azp_dpd_params>
    { // src/DPDPairEvaluatorGeneralWeight.h:32-62
    static azp_dpd_params from_dict(const py::dict& v)
        {
        return azp_dpd_params {v["A"].cast<double>(), v["gamma"].cast<double>(), v["s"].cast<double>(), 0.0};
        }
    static py::dict to_dict(const azp_dpd_params& p)
        {
        py::dict v;
        v["A"] = p.A; v["gamma"] = p.gamma; v["s"] = p.s;
        return v;
        }
    };
template<> struct Codec<azp_tpm_params>
    { // src/AnisoPairEvaluatorTwoPatchMorse.h:32-69
    static azp_tpm_params from_dict(const py::dict& v)
        {
        azp_tpm_params p;
        azp_tpm_params_make(v["M_d"].cast<double>(), v["M_r"].cast<double>(), v["r_eq"].cast<double>(), v["omega"].cast<double>(),
                            v["alpha"].cast<double>(), v["repulsion"].cast<bool>() ? 1 : 0, &p);
        return p;
        }
    static py::dict to_dict(const azp_tpm_params& p)
        {
        double Md, Mr, req, om, al;
        int rep;
        azp_tpm_params_unpack(&p, &Md, &Mr, &req, &om, &al, &rep);
        py::dict v;
        v["M_d"] = Md; v["M_r"] = Mr; v["r_eq"] = req; v["omega"] = om; v["alpha"] = al; v["repulsion"] = (rep != 0);
        return v;
        }
    };
template<> struct Codec<azp_dw_params>
    { // src/BondEvaluatorDoubleWell.h:28-61
    static azp_dw_params from_dict(const py::dict& v)
        {
        azp_dw_params p;
        azp_dw_params_make(v["r_0"].cast<double>(), v["r_1"].cast<double>(), v["U_1"].cast<double>(), v["U_tilt"].cast<double>(), &p);
        return p;
        }
    static py::dict to_dict(const azp_dw_params& p)
        {
        double r0, r1, U1, Ut;
        azp_dw_params_unpack(&p, &r0, &r1, &U1, &Ut);
        py::dict v;
        v["r_0"] = r0; v["r_1"] = r1; v["U_1"] = U1; v["U_tilt"] = Ut;
        return v;
        }
    };
template<> struct Codec<azp_quartic_params>
    { // src/BondEvaluatorQuartic.h:28-82 (delta is optional, default 0: src/bond.py:153)
    static azp_quartic_params from_dict(const py::dict& v)
        {
        azp_quartic_params p;
        const double delta = v.contains("delta") ? v["delta"].cast<double>() : 0.0;
        azp_quartic_params_make(v["k"].cast<double>(), v["r_0"].cast<double>(), v["b_1"].cast<double>(), v["b_2"].cast<double>(),
                                v["U_0"].cast<double>(), v["sigma"].cast<double>(), v["epsilon"].cast<double>(), delta, &p);
        return p;
        }
    static py::dict to_dict(const azp_quartic_params& p)
        {
        double k, r0, b1, b2, U0, s, e, d;
        azp_quartic_params_unpack(&p, &k, &r0, &b1, &b2, &U0, &s, &e, &d);
        py::dict v;
        v["k"] = k; v["r_0"] = r0; v["b_1"] = b1; v["b_2"] = b2; v["U_0"] = U0; v["sigma"] = s; v["epsilon"] = e; v["delta"] = d;
        return v;
        }
    };

class TypeTable
    {
    public:
    explicit TypeTable(std::vector<std::string> types) : m_types(std::move(types))
        {
        if (m_types.empty())
            throw std::runtime_error("at least one type is required");
        }
    size_t index(const std::string& name) const
        {
        for (size_t i = 0; i < m_types.size(); ++i)
            if (m_types[i] == name)
                return i;
        throw std::runtime_error("Type " + name + " not found!"); // HOOMD ParticleData::getTypeByName
        }
    size_t size() const { return m_types.size(); }
    const std::vector<std::string>& names() const { return m_types; }

    private:
    std::vector<std::string> m_types;
    };

// HOOMD PotentialPair<E> / AnisoPotentialPair<E> / PotentialPairDPDThermo<E>: host-side tables
template<class P, int MODES /* bit 0 none, 1 shift, 2 xplor */, bool GPU, int TAG = 0 /* distinct C++ types for classes that share a param_type */>
class PairTables
    {
    public:
    explicit PairTables(const std::vector<std::string>& types)
        : m_types(types), m_params(types.size() * types.size()), m_set(types.size() * types.size(), 0),
          m_rcut(types.size() * types.size(), 0.0), m_ron(types.size() * types.size(), 0.0), m_mode(0)
        {
        std::memset(m_params.data(), 0, sizeof(P) * m_params.size());
        }
    void setParams(const std::string& a, const std::string& b, const py::dict& d)
        {
        const size_t i = m_types.index(a), j = m_types.index(b), T = m_types.size();
        const P p = Codec<P>::from_dict(d);
        m_params[i * T + j] = p; m_params[j * T + i] = p; // HOOMD stores both orderings
        m_set[i * T + j] = m_set[j * T + i] = 1;
        }
    py::dict getParams(const std::string& a, const std::string& b) const
        {
        return Codec<P>::to_dict(m_params[m_types.index(a) * m_types.size() + m_types.index(b)]);
        }
    bool hasParams(const std::string& a, const std::string& b) const { return m_set[m_types.index(a) * m_types.size() + m_types.index(b)] != 0; }
    void setRCut(const std::string& a, const std::string& b, double r)
        {
        const size_t i = m_types.index(a), j = m_types.index(b), T = m_types.size();
        m_rcut[i * T + j] = m_rcut[j * T + i] = r;
        }
    double getRCut(const std::string& a, const std::string& b) const { return m_rcut[m_types.index(a) * m_types.size() + m_types.index(b)]; }
    void setROn(const std::string& a, const std::string& b, double r)
        {
        const size_t i = m_types.index(a), j = m_types.index(b), T = m_types.size();
        m_ron[i * T + j] = m_ron[j * T + i] = r;
        }
    double getROn(const std::string& a, const std::string& b) const { return m_ron[m_types.index(a) * m_types.size() + m_types.index(b)]; }
    void setMode(const std::string& mode)
        {
        const int m = mode == "none" ? 0 : (mode == "shift" ? 1 : (mode == "xplor" ? 2 : -1));
        if (m < 0 || !(MODES & (1 << m)))
            throw std::runtime_error("Invalid energy shift mode: " + mode);
        m_mode = m;
        }
    std::string getMode() const { return m_mode == 0 ? "none" : (m_mode == 1 ? "shift" : "xplor"); }
    int shiftMode() const { return m_mode; }
    py::bytes paramsBytes() const { return py::bytes(reinterpret_cast<const char*>(m_params.data()), sizeof(P) * m_params.size()); }
    std::vector<double> rcutsq() const { return squares(m_rcut); }
    std::vector<double> ronsq() const { return squares(m_ron); }
    std::vector<std::string> types() const { return m_types.names(); }
    static bool onGPU() { return GPU; }
    static size_t paramSize() { return sizeof(P); }

    private:
    static std::vector<double> squares(const std::vector<double>& v)
        {
        std::vector<double> out(v.size());
        for (size_t k = 0; k < v.size(); ++k)
            out[k] = v[k] * v[k];
        return out;
        }
    TypeTable m_types;
    std::vector<P> m_params;
    std::vector<char> m_set;
    std::vector<double> m_rcut, m_ron;
    int m_mode;
    };

// HOOMD PotentialBond<E>: one param_type per bond type
template<class P, bool GPU> class BondTables
    {
    public:
    explicit BondTables(const std::vector<std::string>& types) : m_types(types), m_params(types.size()), m_set(types.size(), 0)
        {
        std::memset(m_params.data(), 0, sizeof(P) * m_params.size());
        }
    void setParams(const std::string& t, const py::dict& d)
        {
        m_params[m_types.index(t)] = Codec<P>::from_dict(d);
        m_set[m_types.index(t)] = 1;
        }
    py::dict getParams(const std::string& t) const { return Codec<P>::to_dict(m_params[m_types.index(t)]); }
    bool hasParams(const std::string& t) const { return m_set[m_types.index(t)] != 0; }
    py::bytes paramsBytes() const { return py::bytes(reinterpret_cast<const char*>(m_params.data()), sizeof(P) * m_params.size()); }
    std::vector<std::string> types() const { return m_types.names(); }
    static bool onGPU() { return GPU; }
    static size_t paramSize() { return sizeof(P); }

    private:
    TypeTable m_types;
    std::vector<P> m_params;
    std::vector<char> m_set;
    };

template<class P, int MODES, bool GPU, int TAG = 0> void export_pair(py::module_& m, const std::string& name)
    {
    typedef PairTables<P, MODES, GPU, TAG> C;
    py::class_<C>(m, name.c_str())
        .def(py::init<const std::vector<std::string>&>(), py::arg("types"))
        .def("setParams", &C::setParams)
        .def("getParams", &C::getParams)
        .def("hasParams", &C::hasParams)
        .def("setRCut", &C::setRCut)
        .def("getRCut", &C::getRCut)
        .def("setROn", &C::setROn)
        .def("getROn", &C::getROn)
        .def_property("mode", &C::getMode, &C::setMode)
        .def_property_readonly("shift_mode", &C::shiftMode)
        .def("params_bytes", &C::paramsBytes)
        .def("rcutsq", &C::rcutsq)
        .def("ronsq", &C::ronsq)
        .def_property_readonly("types", &C::types)
        .def_property_readonly_static("on_gpu", [](py::object) { return C::onGPU(); })
        .def_property_readonly_static("param_size", [](py::object) { return C::paramSize(); });
    }

template<class P, bool GPU> void export_bond(py::module_& m, const std::string& name)
    {
    typedef BondTables<P, GPU> C;
    py::class_<C>(m, name.c_str())
        .def(py::init<const std::vector<std::string>&>(), py::arg("types"))
        .def("setParams", &C::setParams)
        .def("getParams", &C::getParams)
        .def("hasParams", &C::hasParams)
        .def("params_bytes", &C::paramsBytes)
        .def_property_readonly("types", &C::types)
        .def_property_readonly_static("on_gpu", [](py::object) { return C::onGPU(); })
        .def_property_readonly_static("param_size", [](py::object) { return C::paramSize(); });
    }

template<class P, int MODES> void export_pair_both(py::module_& m, const std::string& name)
    {
    export_pair<P, MODES, false>(m, name);
    export_pair<P, MODES, true>(m, name + "GPU");
    }
} // namespace

PYBIND11_MODULE(_azplugins, m)
    {
    m.doc() = "HOOMD-free stand-in for hoomd.azplugins._azplugins (force-compute path): class names of src/module.cc:110-166";
    // bond (src/module.cc:115-116, 141-142)
    export_bond<azp_dw_params, false>(m, "PotentialBondDoubleWell");
    export_bond<azp_dw_params, true>(m, "PotentialBondDoubleWellGPU");
    export_bond<azp_quartic_params, false>(m, "PotentialBondQuartic");
    export_bond<azp_quartic_params, true>(m, "PotentialBondQuarticGPU");
    // pair (src/module.cc:131-135, 154-158): none | shift | xplor; aniso pairs: none | shift
    export_pair_both<azp_tpm_params, 0x3>(m, "AnisoPotentialPairTwoPatchMorse");
    export_pair_both<azp_colloid_params, 0x7>(m, "PotentialPairColloid");
    export_pair_both<azp_yukawa_params, 0x7>(m, "PotentialPairExpandedYukawa");
    export_pair_both<azp_hertz_params, 0x7>(m, "PotentialPairHertz");
    export_pair_both<azp_plj_params, 0x7>(m, "PotentialPairPerturbedLennardJones");
    // dpd pair (src/module.cc:138, 161; src/export_PotentialPairDPDThermo.cc.inc:33-35 also registers the
    // conservative-only PotentialPair<DPDPairEvaluatorGeneralWeight>, CPU class only)
    export_pair<azp_dpd_params, 0x1, false, 1>(m, "PotentialPairConservativeGeneralWeight");
    export_pair_both<azp_dpd_params, 0x1>(m, "PotentialPairDPDThermoGeneralWeight");
    m.attr("azp_version") = azp_version();
    }
