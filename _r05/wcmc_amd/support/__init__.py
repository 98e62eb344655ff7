"""Drop-in mirror of the reference's ``support`` package for the KPCN-Manifold path:
``support.interfaces.KPCNInterface``, ``support.networks.PathNet``, ``support.losses``,
``support.utils.crop_like`` -- same names, signatures and error behaviour; plus the KPCNRef / KPCNPre / SBMC / LBMC
interfaces."""
