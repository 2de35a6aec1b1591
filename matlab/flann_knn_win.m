function [idx, dist] = flann_knn_win(train, query, k, varargin)
    %FLANN_KNN_WIN Replaces the reference's Windows-only mex (PP/mex/flann_knn.cpp) for float descriptors:
    %   exact squared-L2 kNN on the device (method/trees/checks accepted and ignored).
    [idx, dist] = aps_mex('knn_global', single(train), single(query), double(k));
end
